// slab.cpp -- device memory for large feature slabs with a KNOWN address-translation layout.
//
// The aggregation kernels gather 512-byte rows at random from [N, d] matrices of several GB; 56 % of those gathers miss the
// per-CU translation cache (profiles/r03/noreuse_tlb_counters_*: 179 M UTCL1 misses per launch over 8.6 GB), so the page-table
// fragment size of the slab is on the kernel's critical path.  hipMalloc leaves it to the driver; this entry point reserves a
// 1 GiB-aligned virtual range, backs it with ONE physical allocation (hipMemCreate) and maps it in one piece, so that virtual
// and physical alignment allow the largest fragments the VRAM allocator's contiguity permits.
// No reference counterpart: Paddle's allocator owns device memory there (pgl/graph.py:1090-1123, Graph.tensor()).
#include "common.hpp"

#include <mutex>
#include <unordered_map>

namespace {
struct Slab { hipMemGenericAllocationHandle_t handle; size_t bytes; int mode; };
std::mutex g_mu;
std::unordered_map<void*, Slab> g_slabs;
}  // namespace

using namespace pglamd;

extern "C" int32_t pglamd_slab_alloc(size_t bytes, int32_t mode, void** out_ptr, size_t* out_bytes) {
    if (!out_ptr || bytes == 0) return fail(PGLAMD_E_ARG, "slab_alloc: bad argument");
    *out_ptr = nullptr;
    if (mode == 0) {                                   // plain hipMalloc: the comparison partner
        void* p = nullptr;
        PGLAMD_HIP_CHECK(hipMalloc(&p, bytes));
        std::lock_guard<std::mutex> lk(g_mu);
        g_slabs[p] = Slab{nullptr, bytes, 0};
        *out_ptr = p;
        if (out_bytes) *out_bytes = bytes;
        return PGLAMD_OK;
    }
    if (mode != 1) return fail(PGLAMD_E_ARG, "slab_alloc: mode %d (0 = hipMalloc, 1 = one mapped physical allocation in a 1 GiB-aligned range)", mode);
    int dev = 0;
    PGLAMD_HIP_CHECK(hipGetDevice(&dev));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    PGLAMD_HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = (size_t)2 << 20;
    const size_t size = align_up(bytes, gran);
    hipMemGenericAllocationHandle_t h;
    PGLAMD_HIP_CHECK(hipMemCreate(&h, size, &prop, 0));
    void* va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, size, (size_t)1 << 30, nullptr, 0);
    if (e != hipSuccess) { (void)hipMemRelease(h); return fail(PGLAMD_E_HIP, "slab_alloc: hipMemAddressReserve: %s", hipGetErrorString(e)); }
    e = hipMemMap(va, size, 0, h, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(va, size); (void)hipMemRelease(h); return fail(PGLAMD_E_HIP, "slab_alloc: hipMemMap: %s", hipGetErrorString(e)); }
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, size, &acc, 1);
    if (e != hipSuccess) { (void)hipMemUnmap(va, size); (void)hipMemAddressFree(va, size); (void)hipMemRelease(h); return fail(PGLAMD_E_HIP, "slab_alloc: hipMemSetAccess: %s", hipGetErrorString(e)); }
    std::lock_guard<std::mutex> lk(g_mu);
    g_slabs[va] = Slab{h, size, 1};
    *out_ptr = va;
    if (out_bytes) *out_bytes = size;
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_slab_free(void* ptr) {
    if (!ptr) return PGLAMD_OK;
    Slab s;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_slabs.find(ptr);
        if (it == g_slabs.end()) return fail(PGLAMD_E_ARG, "slab_free: %p was not allocated by pglamd_slab_alloc", ptr);
        s = it->second;
        g_slabs.erase(it);
    }
    PGLAMD_HIP_CHECK(hipDeviceSynchronize());
    if (s.mode == 0) { PGLAMD_HIP_CHECK(hipFree(ptr)); return PGLAMD_OK; }
    PGLAMD_HIP_CHECK(hipMemUnmap(ptr, s.bytes));
    PGLAMD_HIP_CHECK(hipMemAddressFree(ptr, s.bytes));
    PGLAMD_HIP_CHECK(hipMemRelease(s.handle));
    return PGLAMD_OK;
}
