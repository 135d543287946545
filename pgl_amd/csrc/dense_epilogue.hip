// dense_epilogue.hip -- the element passes that follow the dense X.W of a conv layer, fused into one row kernel each way
// (row f1, second half: "SpMM -> GEMM epilogue fusion").
//     y[r,:] = normalize_L2( act( z[r,:] + bias ) )                         act: none | relu
// covers GraphSageConv's `self + neigh (+ biases) -> act -> F.normalize` (pgl/nn/conv.py:99-115) and GCNConv's
// `+ bias -> activation` (pgl/nn/conv.py:250-254).  PyTorch runs these as 3-4 kernels forward and 6-8 backward, every one a
// full pass over [N, d] (measured at N = 2^20, d = 128: 45 % of a GraphSage training step); here forward is one read and one
// write of the row, backward one read of (dy, y), one write of dz -- and the bias gradient (column sums of dz) falls out of
// the same pass as per-wave partials.  HBM-bound; lanes across the columns, several rows per wave step, grid-stride over rows.
#include "common.hpp"

namespace pglamd {
namespace {

constexpr int kMaxTiles = 8;           // columns handled per lane = VEC * tiles: d <= 64 * VEC * 8

template <int VEC> struct alignas(4 * VEC) RV { float v[VEC]; };

// sum over the LPR lanes that hold one row (LPR = 64: the whole wave)
template <int LPR> __device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Round 6: rows in flight.  The first form walked ONE row per wave and iteration; at d = 128 (VEC = 4: 32 lanes x 16 B) half the wave idled
// and a CU had 16 KB in flight -- 3.9 TB/s (0.27 ms per [2^20, 128] pass, four to six passes per GraphSage / GCN training step).  Now a
// wave step takes G = 64 / (lanes a row needs) rows side by side (G > 1 only for single-tile rows) and kU such groups are loaded before
// the first is used: d = 128 -> 4 rows = 2 KB per wave in flight.
constexpr int kU = 2;

template <int VEC, int NT, int G>
__global__ __launch_bounds__(kBlock) void row_epilogue_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                              int64_t n_rows, int d, int act, int normalize, float eps,
                                                              float* __restrict__ y, float* __restrict__ inv_norm) {
    static_assert(G == 1 || NT == 1, "several rows per wave step: single-tile rows only");
    using V = RV<VEC>;
    constexpr int LPR = kWave / G;                                   // lanes per row
    const int lane = threadIdx.x & (kWave - 1), sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    V b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = (t * LPR + sl) * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) b[t].v[k] = (bias && j + k < d) ? bias[j + k] : 0.f;
    }
    for (int64_t r0 = wave * (G * kU); r0 < n_rows; r0 += n_waves * (G * kU)) {
        V v[kU][NT];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t r = r0 + u * G + sub;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = (t * LPR + sl) * VEC;
                if (r < n_rows && j < d) v[u][t] = *reinterpret_cast<const V*>(z + r * d + j);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t r = r0 + u * G + sub;
            const bool live = r < n_rows;
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = (t * LPR + sl) * VEC;
                if (live && j < d) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float a = v[u][t].v[k] + b[t].v[k];
                        if (act == 1) a = a > 0.f ? a : 0.f;
                        v[u][t].v[k] = a;
                        ss += a * a;
                    }
                }
            }
            float inv = 1.f;
            if (normalize) {
                ss = row_sum<LPR>(ss);
                const float nrm = sqrtf(ss);
                inv = 1.f / (nrm > eps ? nrm : eps);               // F.normalize: x / max(||x||, eps)
                if (live && sl == 0) inv_norm[r] = inv;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = (t * LPR + sl) * VEC;
                if (live && j < d) {
                    V o;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o.v[k] = v[u][t].v[k] * inv;
                    *reinterpret_cast<V*>(y + r * d + j) = o;
                }
            }
        }
    }
}

// dz = act'(.) * ( normalize ? (dy - y <dy, y>) * inv : dy );  col_part[wave, :] = sum over this wave's rows of dz
template <int VEC, int NT, int G>
__global__ __launch_bounds__(kBlock) void row_epilogue_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  const float* __restrict__ inv_norm, int64_t n_rows, int d, int act,
                                                                  int normalize, float* __restrict__ dz, float* __restrict__ col_part) {
    static_assert(G == 1 || NT == 1, "several rows per wave step: single-tile rows only");
    using V = RV<VEC>;
    constexpr int LPR = kWave / G;
    const int lane = threadIdx.x & (kWave - 1), sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const bool need_y = normalize || act == 1;
    V cs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < VEC; ++k) cs[t].v[k] = 0.f;
    for (int64_t r0 = wave * (G * kU); r0 < n_rows; r0 += n_waves * (G * kU)) {
        V g[kU][NT], yy[kU][NT];
        float inv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t r = r0 + u * G + sub;
            inv[u] = (normalize && r < n_rows) ? inv_norm[r] : 1.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = (t * LPR + sl) * VEC;
                if (r < n_rows && j < d) {
                    g[u][t] = *reinterpret_cast<const V*>(dy + r * d + j);
                    if (need_y) yy[u][t] = *reinterpret_cast<const V*>(y + r * d + j);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t r = r0 + u * G + sub;
            const bool live = r < n_rows;
            float dot = 0.f;
            if (normalize) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int j = (t * LPR + sl) * VEC;
                    if (live && j < d) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) dot += g[u][t].v[k] * yy[u][t].v[k];
                    }
                }
                dot = row_sum<LPR>(dot);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j = (t * LPR + sl) * VEC;
                if (live && j < d) {
                    V o;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float a = g[u][t].v[k];
                        if (normalize) a = (a - yy[u][t].v[k] * dot) * inv[u];
                        if (act == 1) a = yy[u][t].v[k] > 0.f ? a : 0.f;       // y > 0 <=> the pre-activation was > 0
                        o.v[k] = a;
                        cs[t].v[k] += a;
                    }
                    *reinterpret_cast<V*>(dz + r * d + j) = o;
                }
            }
        }
    }
    if (col_part) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = (t * LPR + sl) * VEC;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {                          // the G row groups of the wave hold partial sums of the same columns
                float c = cs[t].v[k];
#pragma unroll
                for (int o = LPR; o < kWave; o <<= 1) c += __shfl_xor(c, o);
                cs[t].v[k] = c;
            }
            if (sub == 0 && j < d) *reinterpret_cast<V*>(col_part + wave * d + j) = cs[t];
        }
    }
}

int grid_blocks(int64_t n_rows) {
    const int64_t want = ceil_div(n_rows, kWavesPerBlock);
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, 256 * 8));      // 8 workgroups per CU, rows by grid stride
}

template <typename F> int32_t dispatch(int d, F&& f) {
    const int vec = d % 4 == 0 ? 4 : d % 2 == 0 ? 2 : 1;
    const int nt = (int)ceil_div(d, (int64_t)kWave * vec);
    if (nt > kMaxTiles) return fail(PGLAMD_E_SHAPE, "row_epilogue: d = %d beyond %d columns", d, kWave * vec * kMaxTiles);
    // rows of at most 32 lanes (d = 128 at VEC = 4): G = 2 / 4 / 8 rows side by side in a wave
    const int lanes = (int)ceil_div(d, vec);
    const int g = nt == 1 ? (lanes <= 8 ? 8 : lanes <= 16 ? 4 : lanes <= 32 ? 2 : 1) : 1;
#define CASE(V, T, G) if (vec == V && nt <= T && g == G) return f(std::integral_constant<int, V>{}, std::integral_constant<int, T>{}, std::integral_constant<int, G>{});
    CASE(4, 1, 8) CASE(4, 1, 4) CASE(4, 1, 2) CASE(2, 1, 8) CASE(2, 1, 4) CASE(2, 1, 2) CASE(1, 1, 8) CASE(1, 1, 4) CASE(1, 1, 2)
    CASE(4, 1, 1) CASE(4, 2, 1) CASE(4, 4, 1) CASE(4, 8, 1) CASE(2, 1, 1) CASE(2, 2, 1) CASE(2, 4, 1) CASE(2, 8, 1) CASE(1, 1, 1) CASE(1, 2, 1) CASE(1, 4, 1) CASE(1, 8, 1)
#undef CASE
    return fail(PGLAMD_E_SHAPE, "row_epilogue: unsupported width %d", d);
}

}  // namespace
}  // namespace pglamd

using namespace pglamd;

extern "C" int64_t pglamd_row_epilogue_partials(int64_t n_rows) { return (int64_t)grid_blocks(n_rows) * kWavesPerBlock; }

extern "C" int32_t pglamd_row_epilogue(const float* z, const float* bias, int64_t n_rows, int64_t d, int32_t act, int32_t normalize,
                                       float eps, float* y, float* inv_norm, void* stream) {
    if (n_rows < 0 || d <= 0 || !z || !y || (normalize && !inv_norm) || act < 0 || act > 1)
        return fail(PGLAMD_E_ARG, "row_epilogue: bad argument");
    if (n_rows == 0) return PGLAMD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dispatch((int)d, [&](auto V, auto T, auto G) -> int32_t {
        hipLaunchKernelGGL((row_epilogue_kernel<decltype(V)::value, decltype(T)::value, decltype(G)::value>), dim3(grid_blocks(n_rows)), dim3(kBlock), 0, st,
                           z, bias, n_rows, (int)d, act, normalize, eps, y, inv_norm);
        PGLAMD_LAUNCH_CHECK();
        return PGLAMD_OK;
    });
}

extern "C" int32_t pglamd_row_epilogue_backward(const float* dy, const float* y, const float* inv_norm, int64_t n_rows, int64_t d,
                                                int32_t act, int32_t normalize, float* dz, float* col_partials, void* stream) {
    if (n_rows < 0 || d <= 0 || !dy || !dz || ((normalize || act == 1) && !y) || (normalize && !inv_norm) || act < 0 || act > 1)
        return fail(PGLAMD_E_ARG, "row_epilogue_backward: bad argument");
    if (n_rows == 0) return PGLAMD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dispatch((int)d, [&](auto V, auto T, auto G) -> int32_t {
        hipLaunchKernelGGL((row_epilogue_bwd_kernel<decltype(V)::value, decltype(T)::value, decltype(G)::value>), dim3(grid_blocks(n_rows)), dim3(kBlock), 0,
                           st, dy, y, inv_norm, n_rows, (int)d, act, normalize, dz, col_partials);
        PGLAMD_LAUNCH_CHECK();
        return PGLAMD_OK;
    });
}
