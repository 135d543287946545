// common.cpp -- error state + version/introspection entry points of libpglamd.
#include "common.hpp"
#include <atomic>

namespace pglamd {

std::string& last_error_ref() {
    thread_local std::string e;
    return e;
}

int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

std::atomic<int>& csr_onesweep_option() {
    static std::atomic<int> v{[] { const char* e = getenv("PGLAMD_CSR_ONESWEEP"); return e ? atoi(e) : -1; }()};
    return v;
}

}  // namespace pglamd

extern "C" int32_t pglamd_abi_version(void) { return PGLAMD_ABI_VERSION; }

extern "C" int32_t pglamd_set_option(const char* name, int64_t value) {
    if (!name) return pglamd::fail(PGLAMD_E_ARG, "set_option: NULL name");
    if (strcmp(name, "csr_onesweep") == 0) { pglamd::csr_onesweep_option().store(value < -1 ? -1 : value > 64 ? 64 : (int)value, std::memory_order_relaxed); return PGLAMD_OK; }
    return pglamd::fail(PGLAMD_E_ARG, "set_option: unknown option %s", name);
}

extern "C" const char* pglamd_last_error(void) { return pglamd::last_error_ref().c_str(); }

extern "C" const char* pglamd_device_arch(void) {
    thread_local std::string arch;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        arch = "";
        return arch.c_str();
    }
    arch = prop.gcnArchName;
    return arch.c_str();
}
