// scan.hpp -- exclusive prefix sum over int64 values produced by a functor (three small coalesced kernels: piece sums -> scan of
// the piece sums in one block -> scan inside the pieces).  Stands where rounds 1-3 called rocprim::exclusive_scan (unique_segment's
// rank of the non-empty rows, the sampler's relabel ranks): the library call brought its own look-back kernels and temporary-storage
// query; the inputs here are flags / degrees of at most a few 10^8 entries.
#pragma once
#include "common.hpp"

namespace pglamd {

constexpr int kScan64Piece = 2048;          // entries per block (256 threads x 8)

__device__ __forceinline__ int64_t block_exclusive_scan64_256(int64_t v, int64_t* wave_tot, int64_t& total) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { const int64_t t = __shfl_up(inc, off, kWave); if (lane >= off) inc += t; }
    __syncthreads();                                    // (wave_tot may still be read by the previous call)
    if (lane == kWave - 1) wave_tot[w] = inc;
    __syncthreads();
    int64_t before = 0, tot = 0;
#pragma unroll
    for (int ww = 0; ww < kBlock / kWave; ++ww) { const int64_t t = wave_tot[ww]; if (ww < w) before += t; tot += t; }
    total = tot;
    return before + inc - v;
}

template <typename F>
__global__ __launch_bounds__(kBlock) void scan64_piece_sums_kernel(F f, int64_t n, int64_t* __restrict__ sums) {
    __shared__ int64_t wave_tot[kBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScan64Piece;
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < kScan64Piece / kBlock; ++i) { const int64_t j = base + i * kBlock + threadIdx.x; if (j < n) s += f(j); }
    int64_t total;
    (void)block_exclusive_scan64_256(s, wave_tot, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

static __global__ __launch_bounds__(kBlock) void scan64_sums_kernel(int64_t* __restrict__ sums, int64_t n) {       // one block
    __shared__ int64_t wave_tot[kBlock / kWave];
    int64_t carry = 0;
    for (int64_t b = 0; b < n; b += kBlock) {
        const int64_t j = b + threadIdx.x;
        const int64_t v = j < n ? sums[j] : 0;
        int64_t total;
        const int64_t ex = block_exclusive_scan64_256(v, wave_tot, total);
        if (j < n) sums[j] = carry + ex;
        carry += total;
    }
}

template <typename F>
__global__ __launch_bounds__(kBlock) void scan64_apply_kernel(F f, int64_t n, const int64_t* __restrict__ sums, int64_t* __restrict__ out) {
    __shared__ int64_t wave_tot[kBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScan64Piece;
    int64_t carry = sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScan64Piece / kBlock; ++i) {
        const int64_t j = base + i * kBlock + threadIdx.x;
        const int64_t v = j < n ? f(j) : 0;
        int64_t total;
        const int64_t ex = block_exclusive_scan64_256(v, wave_tot, total);
        if (j < n) out[j] = carry + ex;
        carry += total;
    }
}

inline size_t exclusive_scan64_temp_bytes(int64_t n) { return align_up((size_t)(ceil_div(n > 0 ? n : 1, (int64_t)kScan64Piece) + 1) * 8, 256); }

// out[i] = sum_{j < i} f(j), i in [0, n).  `out` may alias the array f reads only if f(j) is read before out[j] is written by the
// same thread (it is: apply reads f(j) and writes out[j] in the same iteration) AND no other element is read -- true for the
// elementwise functors used here.  temp: exclusive_scan64_temp_bytes(n).
template <typename F>
int32_t exclusive_scan64(F f, int64_t n, int64_t* out, void* temp, hipStream_t st) {
    if (n <= 0) return PGLAMD_OK;
    int64_t* sums = static_cast<int64_t*>(temp);
    const int64_t pieces = ceil_div(n, (int64_t)kScan64Piece);
    hipLaunchKernelGGL((scan64_piece_sums_kernel<F>), dim3((unsigned)pieces), dim3(kBlock), 0, st, f, n, sums);
    PGLAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan64_sums_kernel, dim3(1), dim3(kBlock), 0, st, sums, pieces);
    PGLAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL((scan64_apply_kernel<F>), dim3((unsigned)pieces), dim3(kBlock), 0, st, f, n, sums, out);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

struct LoadI64 {
    const int64_t* p;
    __device__ int64_t operator()(int64_t i) const { return p[i]; }
};

}  // namespace pglamd
