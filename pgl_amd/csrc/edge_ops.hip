// edge_ops.hip -- K3 send_uv, K4 segment softmax / fused edge_softmax, K5 segment reduce wrapper,
// K6/K7 row gather / scatter, K9 degree_norm, K1' COO atomic scatter-add.
#include "aggregate.hpp"

#include <algorithm>

#include <type_traits>

namespace pglamd {

template <typename T> __device__ __forceinline__ T mop_apply(T a, T b, int mop) {
    switch (mop) {
        case PGLAMD_ADD: return a + b;
        case PGLAMD_SUB: return a - b;
        case PGLAMD_MUL: return a * b;
        default: return a / b;
    }
}

// ------------------------------------------------------------------------------------------------
// K3 send_uv (pgl/graph.py:964-966): out[e, j] = x[src[e], j/gx] (mop) y[dst[e], j/gy]
// One thread per output element group; for the GAT shape (dx=dy=dout=8 fp32) each thread moves a
// float4, two threads per edge, so a wave writes 1 KiB contiguous.
// ------------------------------------------------------------------------------------------------
#ifndef PGLAMD_EDGE_UNROLL           // (variant builds: scripts/prof.py variant NAME PGLAMD_EDGE_UNROLL=4 PGLAMD_EDGE_NT=1)
#define PGLAMD_EDGE_UNROLL 4
#endif
#ifndef PGLAMD_EDGE_NT
#define PGLAMD_EDGE_NT 1
#endif
constexpr int kEdgeUnroll = PGLAMD_EDGE_UNROLL;

// streamed-once operands (edge ids in, [E, ...] results out) bypass the cache hierarchy's retention, so that the 160 MB of
// ids and 640 MB of results of a C3-sized call do not evict the two 32 MB node tables the gathers live on
template <typename V> __device__ __forceinline__ V stream_load(const V* p) {
#if PGLAMD_EDGE_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename V> __device__ __forceinline__ void stream_store(V* p, const V& v) {
#if PGLAMD_EDGE_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// element group i -> (edge, first element of the group): per = groups per edge, a power of two when log2per >= 0
__device__ __forceinline__ void split_group(int64_t i, int64_t per, int log2per, int64_t& e, int64_t& g) {
    if (log2per >= 0) { e = i >> log2per; g = i & (per - 1); }
    else { e = i / per; g = i - e * per; }
}
static int log2_or_neg(int64_t v) { int l = 0; while ((int64_t(1) << l) < v) ++l; return (int64_t(1) << l) == v ? l : -1; }

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void send_uv_vec_kernel(const T* __restrict__ x, const T* __restrict__ y, int64_t d,
                                                             const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                             int64_t E, int mop, int log2per, T* __restrict__ out) {
    typedef T V __attribute__((ext_vector_type(VEC)));
    const int64_t per = d / VEC;
    const int64_t total = E * per;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    // kEdgeUnroll groups per thread and iteration: all their edge ids, then all their gathers are in flight before the first use
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < total; i0 += stride * kEdgeUnroll) {
        int64_t e[kEdgeUnroll], j[kEdgeUnroll];
        int32_t s[kEdgeUnroll], t[kEdgeUnroll];
        V a[kEdgeUnroll], b[kEdgeUnroll];
#pragma unroll
        for (int k = 0; k < kEdgeUnroll; ++k) {
            const int64_t i = i0 + k * stride;
            int64_t g;
            split_group(i < total ? i : 0, per, log2per, e[k], g);
            j[k] = g * VEC;
            s[k] = stream_load(src + e[k]); t[k] = stream_load(dst + e[k]);
        }
#pragma unroll
        for (int k = 0; k < kEdgeUnroll; ++k) {
            a[k] = *reinterpret_cast<const V*>(x + (int64_t)s[k] * d + j[k]);
            b[k] = *reinterpret_cast<const V*>(y + (int64_t)t[k] * d + j[k]);
        }
#pragma unroll
        for (int k = 0; k < kEdgeUnroll; ++k) {
            if (i0 + k * stride >= total) continue;
            V o;
#pragma unroll
            for (int q = 0; q < VEC; ++q) o[q] = mop_apply(a[k][q], b[k][q], mop);
            stream_store(reinterpret_cast<V*>(out + e[k] * d + j[k]), o);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void send_uv_generic_kernel(const T* __restrict__ x, const T* __restrict__ y, int64_t dx,
                                                                 int64_t dy, int64_t dout, const int32_t* __restrict__ src,
                                                                 const int32_t* __restrict__ dst, int64_t E, int mop,
                                                                 T* __restrict__ out) {
    const int64_t gx = dout / dx, gy = dout / dy;
    const int64_t total = E * dout;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t e = i / dout, j = i - e * dout;
        out[i] = mop_apply(x[(int64_t)src[e] * dx + j / gx], y[(int64_t)dst[e] * dy + j / gy], mop);
    }
}

static unsigned grid_for(int64_t n) {
    int64_t g = ceil_div(n > 0 ? n : 1, kBlock);
    return (unsigned)(g < 256 * 32 ? g : 256 * 32);
}

template <typename T>
static int32_t send_uv_typed(const void* x, const void* y, int64_t dx, int64_t dy, int64_t dout, const int32_t* src,
                             const int32_t* dst, int64_t E, int mop, void* out, hipStream_t st) {
    const T* xp = static_cast<const T*>(x); const T* yp = static_cast<const T*>(y); T* op = static_cast<T*>(out);
    constexpr int VEC = 16 / sizeof(T);
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out);
    if (dx == dout && dy == dout && dout % VEC == 0 && al % 16 == 0)
        hipLaunchKernelGGL((send_uv_vec_kernel<T, VEC>), dim3(grid_for(ceil_div(E * (dout / VEC), (int64_t)kEdgeUnroll))), dim3(kBlock), 0, st, xp, yp, dout, src, dst, E, mop, log2_or_neg(dout / VEC), op);
    else
        hipLaunchKernelGGL(send_uv_generic_kernel<T>, dim3(grid_for(E * dout)), dim3(kBlock), 0, st, xp, yp, dx, dy, dout, src, dst, E, mop, op);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

// ------------------------------------------------------------------------------------------------
// K4 segment softmax / edge_softmax.  Load-balanced and deterministic on power-law graphs:
//   m = segment max   -> flat aggregation kernel (K1, reduce = max, gather through `perm`)
//   e = exp(x - m[seg])      -> softmax_exp_kernel, element-parallel in the data's own order
//   s = segment sum of e     -> flat aggregation kernel (reduce = sum)
//   out = e / s[seg]         -> softmax_div_kernel (in place on out)
// Same arithmetic as the reference (pgl/math.py:216-224: e = exp(x - max); e / sum(e)).  With
// perm = sorted_eid and seg = dst the eid gather and the scatter back to original edge order of
// GF.edge_softmax (graph_op.py:117-123) disappear: data is only ever touched in its own order.
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T exp_t(T v);
template <> __device__ __forceinline__ float exp_t<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double exp_t<double>(double v) { return exp(v); }

// MODE 0: exp(x - stat[seg])   1: x / stat[seg]   2: exp(x - max[seg]) / sum[seg] with stat = [n_seg, 2, d] (whole normalisation in one pass)
template <typename T, int VEC, int MODE>
__global__ __launch_bounds__(kBlock) void softmax_elem_kernel(const T* x, const T* __restrict__ stat,
                                                              const int32_t* __restrict__ seg, int64_t n, int64_t d, int log2per,
                                                              T* out) {   // x may alias out (in-place divide)
    typedef T V __attribute__((ext_vector_type(VEC)));
    const int64_t per = d / VEC;
    const int64_t total = n * per;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < total; i0 += stride * kEdgeUnroll) {
        int64_t off[kEdgeUnroll], so[kEdgeUnroll];
        V a[kEdgeUnroll], b[kEdgeUnroll], c[kEdgeUnroll];
#pragma unroll
        for (int k = 0; k < kEdgeUnroll; ++k) {
            const int64_t i = i0 + k * stride;
            int64_t e, g;
            split_group(i < total ? i : 0, per, log2per, e, g);
            off[k] = e * d + g * VEC;
            so[k] = (int64_t)stream_load(seg + e) * (MODE == 2 ? 2 * d : d) + g * VEC;     // MODE 2: [n_seg, 2, d] maxima | sums
            a[k] = stream_load(reinterpret_cast<const V*>(x + off[k]));
        }
#pragma unroll
        for (int k = 0; k < kEdgeUnroll; ++k) {
            b[k] = *reinterpret_cast<const V*>(stat + so[k]);
            if constexpr (MODE == 2) c[k] = *reinterpret_cast<const V*>(stat + so[k] + d);
        }
#pragma unroll
        for (int k = 0; k < kEdgeUnroll; ++k) {
            if (i0 + k * stride >= total) continue;
            V o;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                if constexpr (MODE == 2) o[q] = exp_t<T>(a[k][q] - b[k][q]) / c[k][q];
                else o[q] = MODE == 1 ? a[k][q] / b[k][q] : exp_t<T>(a[k][q] - b[k][q]);
            }
            stream_store(reinterpret_cast<V*>(out + off[k]), o);
        }
    }
}

template <typename T, int MODE>
static int32_t softmax_elem(const T* x, const T* stat, const int32_t* seg, int64_t n, int64_t d, T* out, hipStream_t st) {
    constexpr int VMAX = 16 / sizeof(T);
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(stat) | reinterpret_cast<uintptr_t>(out);
#define SM(VEC)                                                                                                             \
    hipLaunchKernelGGL((softmax_elem_kernel<T, VEC, MODE>), dim3(grid_for(ceil_div(n * (d / VEC), (int64_t)kEdgeUnroll))), \
                       dim3(kBlock), 0, st, x, stat, seg, n, d, log2_or_neg(d / VEC), out)
    if (d % VMAX == 0 && al % 16 == 0) SM(VMAX);
    else if (d % 2 == 0 && al % (2 * sizeof(T)) == 0) SM(2);
    else SM(1);
#undef SM
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

// ------------------------------------------------------------------------------------------------
// K6 / K7 row gather / scatter: pure byte moves, W = widest word that divides the row
// ------------------------------------------------------------------------------------------------
template <typename W, typename I, bool SCATTER>
__global__ __launch_bounds__(kBlock) void move_rows_kernel(const W* __restrict__ x, int64_t row_words, const I* __restrict__ index,
                                                           int64_t n_index, W* __restrict__ out) {
    const int64_t total = n_index * row_words;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / row_words, w = i - r * row_words;
        const int64_t s = (int64_t)index[r];
        if (SCATTER) out[s * row_words + w] = x[i];
        else out[i] = x[s * row_words + w];
    }
}

template <bool SCATTER>
static int32_t move_rows(const void* x, int64_t d, int32_t elem_bytes, const void* index, int32_t index_i64, int64_t n_index,
                         void* out, hipStream_t st) {
    if (n_index == 0 || d == 0) return PGLAMD_OK;
    const size_t row_bytes = (size_t)d * elem_bytes;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out);
#define MV(W)                                                                                                               \
    do {                                                                                                                    \
        const int64_t rw = row_bytes / sizeof(W);                                                                           \
        if (index_i64)                                                                                                      \
            hipLaunchKernelGGL((move_rows_kernel<W, int64_t, SCATTER>), dim3(grid_for(n_index * rw)), dim3(kBlock), 0, st,  \
                               static_cast<const W*>(x), rw, static_cast<const int64_t*>(index), n_index, static_cast<W*>(out)); \
        else                                                                                                                \
            hipLaunchKernelGGL((move_rows_kernel<W, int32_t, SCATTER>), dim3(grid_for(n_index * rw)), dim3(kBlock), 0, st,  \
                               static_cast<const W*>(x), rw, static_cast<const int32_t*>(index), n_index, static_cast<W*>(out)); \
    } while (0)
    if (row_bytes % 16 == 0 && al % 16 == 0) MV(uint4);
    else if (row_bytes % 8 == 0 && al % 8 == 0) MV(uint2);
    else if (row_bytes % 4 == 0 && al % 4 == 0) MV(uint32_t);
    else if (row_bytes % 2 == 0 && al % 2 == 0) MV(uint16_t);
    else MV(uint8_t);
#undef MV
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

// K9 degree_norm (graph_op.py:46-55): pow(max(float(deg), 1), -0.5)
template <typename T>
__global__ __launch_bounds__(kBlock) void degree_norm_kernel(const int64_t* __restrict__ degree, int64_t n, T* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        T v = (T)degree[i];
        v = v < (T)1 ? (T)1 : v;
        out[i] = (T)1 / sqrt(v);
    }
}

// K1' edge-parallel COO scatter-add with hardware fp32 atomics (global_atomic_add_f32): what answers
// paddle.geometric.send_u_recv(x, src, dst) on a SMALL edge list used once (pgl/graph.py:859-861), where the launches of a CSR
// build would cost more than the whole aggregation.  One edge per wave step, lanes across the columns: a wave's 64 atomics hit
// one 256-byte row segment.  Order-nondeterministic in the last bits.
// NOT the path for large graphs: at |E| = 20 M, d = 128 every atomic row is a read-modify-write of a 512-byte line at the memory
// side (the destination matrix does not fit in L2) -- measured 9.96 ms against 0.58 + 1.11 ms for csr_build + the flat kernel
// (profiles/r05/coo.txt).  A variant that walks 8 edges per wave step (ids by one vector load, the 8 row gathers issued back to
// back) was built and measured SLOWER at every size (19.4 ms at C2, 0.041 vs 0.023 ms at 13 k edges: an eighth of the waves, and
// the kernel is bound by the atomics, not by the gathers) and dropped.  Staging destination rows in LDS needs the edges grouped
// by destination tile, i.e. a sorting pass -- which is what csr_build is; so above the crossover (ops.send_u_recv: |E| * d <= 6.4 M
// elements) the engine sorts.
__global__ __launch_bounds__(kBlock) void scatter_add_coo_kernel(const float* __restrict__ x, int64_t d,
                                                                 const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                                 int64_t E, float* __restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t e = wave; e < E; e += nw) {
        const int s = wave_uniform(src[e]);
        const int t = wave_uniform(dst[e]);
        for (int64_t j = lane; j < d; j += kWave) unsafeAtomicAdd(out + (int64_t)t * d + j, x[(int64_t)s * d + j]);
    }
}

}  // namespace pglamd

using namespace pglamd;

extern "C" int32_t pglamd_send_uv(const void* x, const void* y, int32_t dtype, int64_t dx, int64_t dy, int64_t dout,
                                  const int32_t* src, const int32_t* dst, int64_t num_edges, int32_t message_op, void* out,
                                  void* stream) {
    if (num_edges < 0 || (num_edges > 0 && (!x || !y || !src || !dst || !out))) return fail(PGLAMD_E_ARG, "send_uv: bad argument");
    if (dout <= 0 || dx <= 0 || dy <= 0 || dout % dx != 0 || dout % dy != 0)
        return fail(PGLAMD_E_SHAPE, "send_uv: dx=%lld dy=%lld dout=%lld is not a trailing-dim broadcast", (long long)dx, (long long)dy, (long long)dout);
    if (message_op < 0 || message_op > 3) return fail(PGLAMD_E_ARG, "send_uv: bad message_op");
    if (num_edges == 0) return PGLAMD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PGLAMD_F32: return send_uv_typed<float>(x, y, dx, dy, dout, src, dst, num_edges, message_op, out, st);
        case PGLAMD_F64: return send_uv_typed<double>(x, y, dx, dy, dout, src, dst, num_edges, message_op, out, st);
        case PGLAMD_I32: return send_uv_typed<int32_t>(x, y, dx, dy, dout, src, dst, num_edges, message_op, out, st);
        case PGLAMD_I64: return send_uv_typed<int64_t>(x, y, dx, dy, dout, src, dst, num_edges, message_op, out, st);
        default: return fail(PGLAMD_E_DTYPE, "send_uv: dtype %d not supported", dtype);
    }
}

extern "C" size_t pglamd_segment_softmax_workspace_bytes(int64_t num_rows, int64_t d, int64_t n_seg, int32_t dtype) {
    const size_t es = dtype_size(dtype);
    size_t agg = pglamd_aggregate_workspace_bytes(num_rows, d, dtype);
    if (narrow_softmax_covers(d, dtype)) agg = std::max(agg, narrow_softmax_workspace_bytes(num_rows, d, dtype, narrow_chunk_edges()));
    return 2 * align_up((size_t)(n_seg > 0 ? n_seg : 1) * d * es, 256) + agg + 256;
}

extern "C" int32_t pglamd_segment_softmax(const void* data, int32_t dtype, int64_t num_rows, int64_t d, const int32_t* row32,
                                          const int32_t* perm32, const int32_t* seg_of_elem32, const int64_t* seg_ptr,
                                          int64_t n_seg, void* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (num_rows < 0 || n_seg < 0 || d <= 0) return fail(PGLAMD_E_ARG, "segment_softmax: bad size");
    if (num_rows == 0 || n_seg == 0) return PGLAMD_OK;
    if (!data || !out || !seg_ptr || !row32 || !seg_of_elem32) return fail(PGLAMD_E_ARG, "segment_softmax: NULL pointer");
    if (dtype != PGLAMD_F32 && dtype != PGLAMD_F64) return fail(PGLAMD_E_DTYPE, "segment_softmax: dtype %d not supported (F32/F64)", dtype);
    if (!workspace || workspace_bytes < pglamd_segment_softmax_workspace_bytes(num_rows, d, n_seg, dtype))
        return fail(PGLAMD_E_WORKSPACE, "segment_softmax: workspace too small");
    const size_t es = dtype_size(dtype);
    const size_t stat_bytes = align_up((size_t)n_seg * d * es, 256);
    char* w = static_cast<char*>(workspace);
    void* mx = w; void* sm = w + stat_bytes;
    void* aw = w + 2 * stat_bytes;
    const size_t aw_bytes = workspace_bytes - 2 * stat_bytes;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t rc;
    if (narrow_softmax_covers(d, dtype)) {
        // narrow rows: (max, sum of exp) per segment in ONE pass over the data (online softmax, lane per element),
        // then one element-parallel pass writes exp(x - max) / sum.
        rc = narrow_softmax_stats(data, dtype, num_rows, d, row32, perm32, seg_ptr, n_seg, mx, narrow_chunk_edges(), aw, aw_bytes, st);
        if (rc != PGLAMD_OK) return rc;
        if (dtype == PGLAMD_F32)
            return softmax_elem<float, 2>((const float*)data, (const float*)mx, seg_of_elem32, num_rows, d, (float*)out, st);
        return softmax_elem<double, 2>((const double*)data, (const double*)mx, seg_of_elem32, num_rows, d, (double*)out, st);
    }
    rc = pglamd_aggregate(data, dtype, num_rows, d, nullptr, 0, nullptr, row32, perm32, seg_ptr, num_rows, n_seg, n_seg, d,
                          0, PGLAMD_MAX, nullptr, nullptr, 0, mx, aw, aw_bytes, stream);
    if (rc != PGLAMD_OK) return rc;
    if (dtype == PGLAMD_F32) rc = softmax_elem<float, 0>((const float*)data, (const float*)mx, seg_of_elem32, num_rows, d, (float*)out, st);
    else rc = softmax_elem<double, 0>((const double*)data, (const double*)mx, seg_of_elem32, num_rows, d, (double*)out, st);
    if (rc != PGLAMD_OK) return rc;
    rc = pglamd_aggregate(out, dtype, num_rows, d, nullptr, 0, nullptr, row32, perm32, seg_ptr, num_rows, n_seg, n_seg, d, 0,
                          PGLAMD_SUM, nullptr, nullptr, 0, sm, aw, aw_bytes, stream);
    if (rc != PGLAMD_OK) return rc;
    if (dtype == PGLAMD_F32) return softmax_elem<float, 1>((const float*)out, (const float*)sm, seg_of_elem32, num_rows, d, (float*)out, st);
    return softmax_elem<double, 1>((const double*)out, (const double*)sm, seg_of_elem32, num_rows, d, (double*)out, st);
}

extern "C" size_t pglamd_segment_reduce_workspace_bytes(int64_t num_rows, int64_t d, int64_t n_out_rows, int32_t dtype) {
    const int64_t n = num_rows > 0 ? num_rows : 1;
    // seg_ptr + ids32 (int64 ids are narrowed) + partials of the flat aggregation
    return align_up((size_t)(n_out_rows + 1) * 8, 256) + align_up((size_t)n * 4, 256) +
           pglamd_aggregate_workspace_bytes(num_rows, d, dtype) + 512;
}

extern "C" int32_t pglamd_segment_reduce(const void* data, int32_t dtype, const void* ids, int32_t ids_i64, int64_t num_rows,
                                         int64_t d, int64_t n_out_rows, int32_t reduce_op, void* out, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    if (num_rows < 0 || n_out_rows < 0 || d <= 0) return fail(PGLAMD_E_ARG, "segment_reduce: bad size");
    if (n_out_rows == 0) return PGLAMD_OK;
    if (!out || (num_rows > 0 && (!data || !ids))) return fail(PGLAMD_E_ARG, "segment_reduce: NULL pointer");
    const size_t seg_bytes = align_up((size_t)(n_out_rows + 1) * 8, 256);
    const size_t id_bytes = ids_i64 ? align_up((size_t)(num_rows > 0 ? num_rows : 1) * 4, 256) : 0;
    const size_t agg_bytes = pglamd_aggregate_workspace_bytes(num_rows, d, dtype);
    if (!workspace || workspace_bytes < seg_bytes + id_bytes + agg_bytes)
        return fail(PGLAMD_E_WORKSPACE, "segment_reduce: workspace %zu < %zu (+ (n_out_rows+1)*8 for seg_ptr)", workspace_bytes,
                    seg_bytes + id_bytes + agg_bytes);
    char* w = static_cast<char*>(workspace);
    int64_t* seg_ptr = reinterpret_cast<int64_t*>(w); w += seg_bytes;
    const int32_t* ids32 = static_cast<const int32_t*>(ids);
    if (ids_i64) {
        int32_t* tmp = reinterpret_cast<int32_t*>(w); w += id_bytes;
        int32_t rc = pglamd_narrow_i64(static_cast<const int64_t*>(ids), 1, num_rows, tmp, stream);
        if (rc != PGLAMD_OK) return rc;
        ids32 = tmp;
    }
    int32_t rc = pglamd_seg_ptr_from_ids(ids32, 0, num_rows, n_out_rows, seg_ptr, stream);
    if (rc != PGLAMD_OK) return rc;
    return pglamd_aggregate(data, dtype, num_rows, d, nullptr, 0, nullptr, ids32, nullptr, seg_ptr, num_rows, n_out_rows,
                            n_out_rows, d, 0, reduce_op, nullptr, nullptr, 0, out, w, workspace_bytes - (size_t)(w - static_cast<char*>(workspace)),
                            stream);
}

extern "C" int32_t pglamd_gather_rows(const void* x, int64_t d, int32_t elem_bytes, const void* index, int32_t index_i64,
                                      int64_t n_index, void* out, void* stream) {
    if (n_index < 0 || d < 0 || (n_index > 0 && d > 0 && (!x || !index || !out))) return fail(PGLAMD_E_ARG, "gather_rows: bad argument");
    return move_rows<false>(x, d, elem_bytes, index, index_i64, n_index, out, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------
// K6w  row gather with a dtype change: the wire pack / unpack of the halo exchange (pgl_amd.distributed, 16-bit wire for
// fp32 features).  out[i, :] = cast(x[index[i], :])  (index NULL: identity = a plain row-wise conversion).
// One thread moves 4 elements (16 bytes of fp32 in or out), rows are contiguous.
// ------------------------------------------------------------------------------------------------
namespace {
template <typename T> __device__ __forceinline__ float wire_to_f32(T v);
template <> __device__ __forceinline__ float wire_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float wire_to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float wire_to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T wire_from_f32(float v);
template <> __device__ __forceinline__ float wire_from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half wire_from_f32<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ __hip_bfloat16 wire_from_f32<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename TI, typename TO, int VEC>
__global__ __launch_bounds__(kBlock) void gather_cast_kernel(const TI* __restrict__ x, int64_t d, int64_t ldx, const int32_t* __restrict__ index,
                                                             int64_t n_index, TO* __restrict__ out) {
    struct alignas(sizeof(TI) * VEC) VI { TI v[VEC]; };
    struct alignas(sizeof(TO) * VEC) VO { TO v[VEC]; };
    const int64_t per = d / VEC, total = n_index * per;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / per, j = (i - r * per) * VEC;
        const int64_t s = index ? (int64_t)index[r] : r;
        const VI a = *reinterpret_cast<const VI*>(x + s * ldx + j);
        VO o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = wire_from_f32<TO>(wire_to_f32<TI>(a.v[k]));
        *reinterpret_cast<VO*>(out + r * d + j) = o;
    }
}

template <typename TI, typename TO>
int32_t gather_cast(const void* x, int64_t d, int64_t ldx, const int32_t* index, int64_t n, void* out, hipStream_t st) {
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out);
    if (d % 4 == 0 && ldx % 4 == 0 && al % 16 == 0)
        hipLaunchKernelGGL((gather_cast_kernel<TI, TO, 4>), dim3(grid_for(n * (d / 4))), dim3(kBlock), 0, st, static_cast<const TI*>(x), d, ldx, index, n, static_cast<TO*>(out));
    else
        hipLaunchKernelGGL((gather_cast_kernel<TI, TO, 1>), dim3(grid_for(n * d)), dim3(kBlock), 0, st, static_cast<const TI*>(x), d, ldx, index, n, static_cast<TO*>(out));
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}
}  // namespace

extern "C" int32_t pglamd_gather_rows_cast(const void* x, int32_t x_dtype, int64_t d, int64_t ldx, const int32_t* index, int64_t n_index,
                                           void* out, int32_t out_dtype, void* stream) {
    if (n_index < 0 || d < 0 || (n_index > 0 && d > 0 && (!x || !out))) return fail(PGLAMD_E_ARG, "gather_rows_cast: bad argument");
    if (ldx == 0) ldx = d;
    if (ldx < d) return fail(PGLAMD_E_SHAPE, "gather_rows_cast: row stride %lld shorter than the row %lld", (long long)ldx, (long long)d);
    if (n_index == 0 || d == 0) return PGLAMD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (x_dtype == PGLAMD_F32 && out_dtype == PGLAMD_F16) return gather_cast<float, __half>(x, d, ldx, index, n_index, out, st);
    if (x_dtype == PGLAMD_F32 && out_dtype == PGLAMD_BF16) return gather_cast<float, __hip_bfloat16>(x, d, ldx, index, n_index, out, st);
    if (x_dtype == PGLAMD_F16 && out_dtype == PGLAMD_F32) return gather_cast<__half, float>(x, d, ldx, index, n_index, out, st);
    if (x_dtype == PGLAMD_BF16 && out_dtype == PGLAMD_F32) return gather_cast<__hip_bfloat16, float>(x, d, ldx, index, n_index, out, st);
    if (x_dtype == PGLAMD_F32 && out_dtype == PGLAMD_F32) return gather_cast<float, float>(x, d, ldx, index, n_index, out, st);
    if (x_dtype == PGLAMD_F16 && out_dtype == PGLAMD_F16) return gather_cast<__half, __half>(x, d, ldx, index, n_index, out, st);
    if (x_dtype == PGLAMD_BF16 && out_dtype == PGLAMD_BF16) return gather_cast<__hip_bfloat16, __hip_bfloat16>(x, d, ldx, index, n_index, out, st);
    return fail(PGLAMD_E_DTYPE, "gather_rows_cast: F32 <-> F16 / BF16, or the same 16- / 32-bit float type (got %d -> %d)", x_dtype, out_dtype);
}

extern "C" int32_t pglamd_scatter_rows(const void* x, int64_t d, int32_t elem_bytes, const void* index, int32_t index_i64,
                                       int64_t n_index, void* out, void* stream) {
    if (n_index < 0 || d < 0 || (n_index > 0 && d > 0 && (!x || !index || !out))) return fail(PGLAMD_E_ARG, "scatter_rows: bad argument");
    return move_rows<true>(x, d, elem_bytes, index, index_i64, n_index, out, static_cast<hipStream_t>(stream));
}

extern "C" int32_t pglamd_degree_norm(const int64_t* degree, int64_t n, void* out, int32_t out_f64, void* stream) {
    if (n < 0 || (n > 0 && (!degree || !out))) return fail(PGLAMD_E_ARG, "degree_norm: bad argument");
    if (n == 0) return PGLAMD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out_f64) hipLaunchKernelGGL(degree_norm_kernel<double>, dim3(grid_for(n)), dim3(kBlock), 0, st, degree, n, static_cast<double*>(out));
    else hipLaunchKernelGGL(degree_norm_kernel<float>, dim3(grid_for(n)), dim3(kBlock), 0, st, degree, n, static_cast<float*>(out));
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_scatter_add_coo(const float* x, int64_t d, const int32_t* src, const int32_t* dst, int64_t num_edges,
                                          int64_t out_rows, float* out, void* stream) {
    if (num_edges < 0 || out_rows < 0 || d <= 0 || !out || (num_edges > 0 && (!x || !src || !dst)))
        return fail(PGLAMD_E_ARG, "scatter_add_coo: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    PGLAMD_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)out_rows * d * sizeof(float), st));
    if (num_edges == 0) return PGLAMD_OK;
    int64_t g = ceil_div(num_edges, kWavesPerBlock);
    hipLaunchKernelGGL(scatter_add_coo_kernel, dim3((unsigned)(g < 256 * 32 ? g : 256 * 32)), dim3(kBlock), 0, st, x, d, src, dst,
                       num_edges, out);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}
