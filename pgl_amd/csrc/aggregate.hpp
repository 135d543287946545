// aggregate.hpp -- launch parameters and device helpers shared by the aggregation kernels
// (aggregate.hip: lanes across the feature dimension; aggregate_narrow.hip: one lane per edge).
#pragma once
#include <atomic>
#include <mutex>
#include "common.hpp"

#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace pglamd {

struct AggParams {
    const void* x; const void* y; void* out;
    const void* x2;                       // second source table, ALREADY rebased: source row c >= x_split is read at x2 + c * ldx
    const int64_t* zero_indptr;           // rows r with zero_indptr[r] == zero_indptr[r+1] are the ones the zero-fill role clears
    const int* row; const int* col; const int* eid;
    const int64_t* indptr;
    const float* src_scale; const float* dst_scale;
    int ss_by_pos;                        // 1: src_scale holds one value per EDGE POSITION of the sorted stream (src_scale[p], read in order) instead of one per source node
    void* part_head; void* part_tail;     // [n_chunks, tile_cols] of ACC each
    int* long_count; int* long_list;      // [2] counters + work list of split-row fix-up tasks (workspace)
    int* long_list2;                      // second-level list: rows with more than kFixShort partials
    int64_t ldx, ldy, ldo;                // row strides (elements) of x, y, out
    int64_t out_rows, n_csr_rows;
    int E, n_chunks, chunk, n_blocks;
    int n_grid_chunks;                    // blocks [0, n_grid_chunks) walk edge chunks, the rest zero-fill
    int j_base, tile_cols;                // this launch covers out columns [j_base, j_base+tile_cols)
    int gy;                               // y column = j / gy   (YMODE 1 / 3)
    int ypad;                             // YMODE 3: y row length rounded up to a power of two (<= 8); 0 = not applicable
    int mop, is_max, is_mean;
    int zvec;                             // vector width the zero-fill role may use (1, 2, 4)
    int accumulate;                       // 0: write every row; 1: combine rows that receive edges with their old contents; 2: overwrite only those rows
    int align;                            // 1: never split rows of <= chunk edges (chunk_cut)
    int narrow_vec;                       // aggregate_narrow: rows may be moved with (<=16-byte) vector loads / stores
    int x_split;                          // INT32_MAX when there is no second table
    int deal_chunks;                      // 1: logical chunk blocks are dealt round the XCDs (launch_flat) instead of in contiguous runs
    // ---- SINK = 1 (aggregate_flat.hpp): the aggregated rows feed a dense layer without leaving the chip -------------------------
    const float* w;                       // [d_in, dout2] row-major weight (split-row fix-up: one matrix-vector product per hub row)
    const float* wp;                      // the same weight packed in MFMA B-operand order: wp[(ct * (d_in / 4) + kk) * 64 + lane]
    const float* bias;                    // [dout2] or NULL
    float* out2;                          // [out_rows, dout2]: act(dst_scale * aggregate(x) @ w + bias)
    int dout2, act;                       // act: 0 none, 1 relu
    int max_row_edges;                    // host-side hint: longest row of the index (0 = unknown).  Rows of <= chunk edges are never
                                          // split (chunk_cut), so when it is <= chunk no partial exists and the fix-up launches are skipped
};

// true when this launch can leave split-row partials behind (=> the counter reset and the two fix-up launches are needed)
inline bool needs_fixups(const AggParams& p) {
    return p.n_chunks > 1 && !(p.align && p.max_row_edges > 0 && p.max_row_edges <= p.chunk);
}

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) VecT { T v[VEC]; };

// storage type T -> accumulator type A: 16-bit floats are accumulated (and their partials kept) in fp32
template <typename T> struct AccT { using type = T; };
template <> struct AccT<__half> { using type = float; };
template <> struct AccT<__hip_bfloat16> { using type = float; };
template <typename T> __device__ __forceinline__ typename AccT<T>::type to_acc(T v) { return v; }
template <> __device__ __forceinline__ float to_acc<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_acc<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_acc(typename AccT<T>::type v) { return v; }
template <> __device__ __forceinline__ __half from_acc<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_acc<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename T> struct Limits;
template <> struct Limits<float> { static __device__ float lo() { return -INFINITY; } static __device__ float hi() { return INFINITY; } };
template <> struct Limits<double> { static __device__ double lo() { return -INFINITY; } static __device__ double hi() { return INFINITY; } };
template <> struct Limits<int32_t> { static __device__ int32_t lo() { return INT32_MIN; } static __device__ int32_t hi() { return INT32_MAX; } };
template <> struct Limits<int64_t> { static __device__ int64_t lo() { return INT64_MIN; } static __device__ int64_t hi() { return INT64_MAX; } };

template <typename T> __device__ __forceinline__ T apply_mop(T a, T b, int mop) {
    switch (mop) {
        case PGLAMD_ADD: return a + b;
        case PGLAMD_SUB: return a - b;
        case PGLAMD_MUL: return a * b;
        default: return a / b;
    }
}

// Zero-fills the columns [j_base, j_base+tile_cols) of output rows that receive no edge: rows
// r < n_csr_rows with indptr[r]==indptr[r+1], and rows in [n_csr_rows, out_rows).  One wave
// inspects 64 rows (coalesced indptr read) and clears the empty ones, lanes across the columns.
template <typename T>
__device__ __forceinline__ void zero_empty_rows_role(const AggParams& p, int64_t zb, int lane) {
    const int64_t w = zb * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t r0 = w * kWave;
    if (r0 >= p.out_rows) return;
    const int64_t r = r0 + lane;
    bool empty = false;
    if (r < p.out_rows) empty = (r >= p.n_csr_rows) || (p.zero_indptr[r] == p.zero_indptr[r + 1]);
    unsigned long long m = __ballot(empty);
    T* out = static_cast<T*>(p.out) + p.j_base;
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        T* dst = out + (r0 + l) * p.ldo;
        if (p.zvec == 4) {
            for (int j = lane * 4; j < p.tile_cols; j += kWave * 4) *reinterpret_cast<VecT<T, 4>*>(dst + j) = VecT<T, 4>{};
        } else if (p.zvec == 2) {
            for (int j = lane * 2; j < p.tile_cols; j += kWave * 2) *reinterpret_cast<VecT<T, 2>*>(dst + j) = VecT<T, 2>{};
        } else {
            for (int j = lane; j < p.tile_cols; j += kWave) dst[j] = from_acc<T>(typename AccT<T>::type(0));
        }
    }
}

// Dense sink: an output row without any edge aggregates to 0, so its layer output is act(0 @ w + bias) = act(bias).
__device__ __forceinline__ void dense_empty_rows_role(const AggParams& p, int64_t zb, int lane) {
    const int64_t w = zb * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t r0 = w * kWave;
    if (r0 >= p.out_rows) return;
    const int64_t r = r0 + lane;
    bool empty = false;
    if (r < p.out_rows) empty = (r >= p.n_csr_rows) || (p.zero_indptr[r] == p.zero_indptr[r + 1]);
    unsigned long long m = __ballot(empty);
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        float* dst = p.out2 + (r0 + l) * (int64_t)p.dout2;
        for (int j = lane; j < p.dout2; j += kWave) {
            float v = p.bias ? p.bias[j] : 0.f;
            if (p.act) v = v > 0.f ? v : 0.f;
            dst[j] = v;
        }
    }
}

// Optional in-library timing of the aggregation kernels (bench.py roofline leg): while enabled, every edge-kernel launch is
// bracketed by a pair of HIP events on the launch stream.
struct ProfileState {
    std::atomic<bool> on{false};     // read on every launch; everything else only under `mu` and only while on
    std::mutex mu;                   // autograd backward threads launch concurrently with the main thread
    std::string last_kernel;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};
ProfileState& prof();      // defined once, in aggregate.hip

// aggregate.hip: edges per chunk (256; PGLAMD_CHUNK overrides, used by the stress tests)
int chunk_edges();
int chunk_edges_for(int64_t num_edges);     // size-aware default (aggregate.hip)
int narrow_chunk_edges();
// aggregate.hip: combines the T[a] / H[c] partials the edge kernels leave for rows longer than a chunk
// (tile_cols <= 64 columns, one column per lane).
int32_t launch_fixup_cols(const AggParams& p, int32_t dtype, int rcls, hipStream_t st);
// aggregate_narrow.hip: rows of <= 16 elements, one lane per edge.  *handled = false when the shape is not covered.
int32_t launch_narrow(const AggParams& p, int32_t dtype, int rcls, int64_t dy, hipStream_t st, bool* handled);

// aggregate_narrow.hip: one-pass (online) softmax statistics for rows of <= 16 fp32 / <= 8 fp64 elements.
bool narrow_softmax_covers(int64_t d, int32_t dtype);
size_t narrow_softmax_workspace_bytes(int64_t num_rows, int64_t d, int32_t dtype, int chunk);
int32_t narrow_softmax_stats(const void* data, int32_t dtype, int64_t num_rows, int64_t d, const int32_t* row32,
                             const int32_t* perm32, const int64_t* seg_ptr, int64_t n_seg, void* stats,
                             int chunk, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace pglamd
