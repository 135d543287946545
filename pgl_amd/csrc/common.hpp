// common.hpp -- shared host/device helpers for libpglamd (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

#include <atomic>
#include <cstdarg>
#include <cstring>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/pgl_amd.h"

namespace pglamd {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kXcds = 8;           // MI355X: block b is dispatched to XCD b % 8
// Edge positions are 32-bit inside the kernels and chunk ends are computed as c * K + K before clamping, so the
// engine accepts edge counts that leave that headroom (per graph / per partition).
constexpr int64_t kMaxEdges = (int64_t)INT32_MAX - 65536;

std::string& last_error_ref();
int32_t fail(int32_t code, const char* fmt, ...);

#define PGLAMD_HIP_CHECK(expr)                                                             \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ::pglamd::fail(PGLAMD_E_HIP, "%s failed: %s (%s:%d)", #expr,            \
                                  hipGetErrorString(_e), __FILE__, __LINE__);              \
    } while (0)

#define PGLAMD_LAUNCH_CHECK() PGLAMD_HIP_CHECK(hipGetLastError())

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline size_t dtype_size(int32_t dt) {
    switch (dt) {
        case PGLAMD_F16: case PGLAMD_BF16: return 2;
        case PGLAMD_F32: case PGLAMD_I32: return 4;
        case PGLAMD_F64: case PGLAMD_I64: return 8;
        default: return 0;
    }
}

// Bump allocator over the caller's workspace (256-byte aligned slices).
struct Carver {
    char* base; size_t cap; size_t off = 0;
    Carver(void* p, size_t n) : base(static_cast<char*>(p)), cap(n) {}
    template <typename T> T* take(size_t count) {
        size_t bytes = align_up(count * sizeof(T), 256);
        T* r = reinterpret_cast<T*>(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return base != nullptr && off <= cap; }
};

// blockIdx -> logical block so that CONSECUTIVE logical blocks land on the SAME XCD
// (dispatcher places block b on XCD b % 8; speed only, never correctness).
// Launch ceil(nb/8)*8 blocks; returns -1 for the padding blocks.
__device__ __forceinline__ int64_t xcd_swizzle(int64_t b, int64_t nb) {
    if (nb < 0) return b < -nb ? b : -1;       // -nb blocks, NOT remapped: block b is chunk b, i.e. consecutive chunks go round the XCDs
    int64_t per = (nb + kXcds - 1) / kXcds;
    int64_t lb = (b % kXcds) * per + b / kXcds;
    return lb < nb ? lb : -1;
}
inline int64_t xcd_grid(int64_t nb) { return ceil_div(nb, kXcds) * kXcds; }
std::atomic<int>& csr_onesweep_option();       // common.cpp: csr_build.hip sweep_group (pglamd_set_option "csr_onesweep")

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Read-only kernel inputs addressed wave-uniformly (index arrays, per-row scalars) are viewed
// through the CONSTANT address space: the backend then always emits scalar loads
// (s_load_dword..x8 through the scalar cache) instead of a vector load + v_readfirstlane.
// Only valid for memory no thread of the running kernel writes.
template <typename T> using cptr = const T __attribute__((address_space(4)))*;
template <typename T> __device__ __forceinline__ cptr<T> as_const(const T* p) {
    return (cptr<T>)(unsigned long long)p;
}

// Edge range [e0, e1) of chunk c of a dst-sorted edge stream cut on the grid c*K.  A row with <= K
// edges is never split: it belongs wholly to the chunk in which it STARTS (so a chunk walks between
// 0 and 2K-1 edges).  Only rows longer than K are cut at grid positions; those leave partials.
__device__ __forceinline__ int chunk_cut(cptr<int> rowp, cptr<int64_t> ip, int pos, int K, int E) {
    if (pos <= 0) return 0;
    if (pos >= E) return E;
    const int r = rowp[pos];
    const int64_t rs = ip[r];
    if (rs == pos) return pos;                 // grid position is a row start
    const int64_t re = ip[r + 1];
    if (re - rs > K) return pos;               // long row: cut here
    return (int)re;                            // short row started in an earlier chunk: skip past it
}

}  // namespace pglamd
