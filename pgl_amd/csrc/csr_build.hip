// csr_build.hip -- K8: COO -> CSR on the GPU, bit-identical to the reference's build_index
// (pgl/graph_kernel.pyx:59-88: stable counting sort by key, ascending original edge id inside a
// row), plus unique_segment (pgl/utils/helper.py:156-160) and the seg_ptr helper.
//
// Plan (all HBM-bound integer work, ~28 B/edge algorithmic):
//   1. narrow the int64 keys (strided: a column of the [E,2] edge array) to int32, eid = iota;
//   2. stable LSD radix sort of (key32, eid32) over ceil(log2 N) key bits only
//      (rocPRIM device radix sort, the AMD-native primitive: onesweep passes tuned for gfx9);
//      stability == ascending eid inside equal keys == the reference's order by construction;
//   3. indptr from row boundaries in the sorted keys (no atomics, no scan): every position p with
//      key[p] != key[p-1] writes indptr for the rows in (key[p-1], key[p]];  degree = diff;
//   4. the neighbour id v rides THROUGH the sort packed with the edge id in one 64-bit value (v32 << 32 | eid32): 4 more
//      bytes per element per pass, all sequential, instead of a random 8-byte gather v[eid] per edge afterwards (one
//      128-byte line per edge: 0.40 ms of the 1.04 at C2); the last kernel unpacks sequentially, widens to the
//      reference's int64 arrays where asked, and keeps the int32 (row, col, eid) copies the aggregation kernels read.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace pglamd {

static int key_bits(int64_t n) {
    int b = 1;
    while (b < 32 && (int64_t(1) << b) < n) ++b;
    return b;
}

__global__ __launch_bounds__(kBlock) void narrow_keys_kernel(const int64_t* __restrict__ u, int64_t u_stride, const int64_t* __restrict__ v,
                                                             int64_t v_stride, int64_t n, int64_t n_rows, int32_t* __restrict__ key,
                                                             uint64_t* __restrict__ val, int32_t* __restrict__ range_flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        int64_t k = u[i * u_stride];
        const int64_t nb = v[i * v_stride];
        // a key outside [0, N) would be sorted on its low bits only and then race in row_bounds_kernel; a neighbour that
        // does not fit 31 bits would alias another node.  Both are clamped (memory-safe) and reported through range_flag.
        if ((uint64_t)k >= (uint64_t)n_rows) { k = 0; bad = true; }
        if ((uint64_t)nb > (uint64_t)INT32_MAX) bad = true;
        key[i] = (int32_t)k;
        val[i] = ((uint64_t)(uint32_t)nb << 32) | (uint64_t)(uint32_t)i;       // (neighbour, original edge id)
    }
    if (range_flag && __any(bad)) { if ((threadIdx.x & (kWave - 1)) == 0) atomicOr(range_flag, 1); }
}

__global__ __launch_bounds__(kBlock) void narrow_i64_kernel(const int64_t* __restrict__ in, int64_t stride, int64_t n, int32_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (int32_t)in[i * stride];
}

// indptr[r] = first position whose key >= r  (keys sorted); also covers r in (last key, n_rows]
template <typename K>
__global__ __launch_bounds__(kBlock) void row_bounds_kernel(const K* __restrict__ key, int64_t n, int64_t n_rows,
                                                            int64_t* __restrict__ indptr) {
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p <= n; p += (int64_t)gridDim.x * kBlock) {
        const int64_t prev = p == 0 ? -1 : (int64_t)key[p - 1];
        int64_t cur = p == n ? n_rows : (int64_t)key[p];
        if (cur > n_rows) cur = n_rows;
        for (int64_t r = prev + 1; r <= cur; ++r) indptr[r] = p;
    }
}

__global__ __launch_bounds__(kBlock) void finish_csr_kernel(const int32_t* __restrict__ row32, const uint64_t* __restrict__ val, int64_t n,
                                                            int64_t* __restrict__ sorted_v, int64_t* __restrict__ sorted_u,
                                                            int64_t* __restrict__ sorted_eid, int32_t* __restrict__ col32,
                                                            int32_t* __restrict__ eid32) {
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const uint64_t w = val[p];
        const int32_t e = (int32_t)(uint32_t)w, vv = (int32_t)(uint32_t)(w >> 32);
        if (sorted_v) sorted_v[p] = vv;
        if (sorted_u) sorted_u[p] = row32[p];
        if (sorted_eid) sorted_eid[p] = e;
        if (col32) col32[p] = vv;
        if (eid32) eid32[p] = e;
    }
}

__global__ __launch_bounds__(kBlock) void degree_kernel(const int64_t* __restrict__ indptr, int64_t n_rows,
                                                        int64_t* __restrict__ degree) {
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * kBlock)
        degree[r] = indptr[r + 1] - indptr[r];
}

struct NonEmpty {
    const int64_t* degree;
    __device__ int64_t operator()(int64_t r) const { return degree[r] > 0 ? 1 : 0; }
};

__global__ __launch_bounds__(kBlock) void uniq_scatter_kernel(const int64_t* __restrict__ degree, const int64_t* __restrict__ rank,
                                                              int64_t n_rows, int64_t* __restrict__ uniq, int64_t* __restrict__ num_uniq) {
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * kBlock) {
        if (degree[r] > 0) uniq[rank[r]] = r;
        if (r == n_rows - 1) *num_uniq = rank[r] + (degree[r] > 0 ? 1 : 0);
    }
}

__global__ __launch_bounds__(kBlock) void seg_ids_kernel(const int64_t* __restrict__ sorted_u, const int64_t* __restrict__ rank,
                                                         int64_t n, int64_t* __restrict__ seg) {
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock)
        seg[p] = rank[sorted_u[p]];
}

static unsigned grid_for(int64_t n) {
    int64_t g = ceil_div(n > 0 ? n : 1, kBlock);
    return (unsigned)(g < 256 * 16 ? g : 256 * 16);
}

static size_t sort_temp_bytes(int64_t E, int bits) {
    size_t bytes = 0;
    int32_t* k = nullptr;
    uint64_t* w = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, w, w, (size_t)(E > 0 ? E : 1), 0u, (unsigned)bits, (hipStream_t)0);
    return bytes;
}

static size_t scan_temp_bytes(int64_t N) {
    size_t bytes = 0;
    auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), NonEmpty{nullptr});
    int64_t* o = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, it, o, int64_t(0), (size_t)(N > 0 ? N : 1), rocprim::plus<int64_t>(), (hipStream_t)0);
    return bytes;
}

}  // namespace pglamd

using namespace pglamd;

extern "C" size_t pglamd_csr_build_workspace_bytes(int64_t num_edges, int64_t num_nodes) {
    const int64_t E = num_edges > 0 ? num_edges : 1;
    // key_in, row32 (when the caller does not keep it), packed values in / out, sort temp
    return 2 * align_up((size_t)E * 4, 256) + 2 * align_up((size_t)E * 8, 256) + align_up(sort_temp_bytes(E, key_bits(num_nodes)), 256) + 1024;
}

extern "C" int32_t pglamd_csr_build(const int64_t* u, int64_t u_stride, const int64_t* v, int64_t v_stride,
                                    int64_t num_edges, int64_t num_nodes, int64_t* degree, int64_t* sorted_v,
                                    int64_t* sorted_u, int64_t* sorted_eid, int64_t* indptr, int32_t* row32,
                                    int32_t* col32, int32_t* eid32, int32_t* range_flag, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (num_edges < 0 || num_nodes < 0 || num_edges >= INT32_MAX || num_nodes >= INT32_MAX)
        return fail(PGLAMD_E_RANGE, "csr_build: E=%lld N=%lld beyond the int32 engine range", (long long)num_edges, (long long)num_nodes);
    if (!indptr || (num_nodes > 0 && !degree) || (num_edges > 0 && (!u || !v))) return fail(PGLAMD_E_ARG, "csr_build: NULL pointer");   // (an empty degree array has no address)
    if (!workspace || workspace_bytes < pglamd_csr_build_workspace_bytes(num_edges, num_nodes))
        return fail(PGLAMD_E_WORKSPACE, "csr_build: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t E = num_edges, N = num_nodes;
    Carver cv(workspace, workspace_bytes);
    int32_t* key_in = cv.take<int32_t>(E > 0 ? E : 1);
    int32_t* row_tmp = cv.take<int32_t>(E > 0 ? E : 1);
    uint64_t* val_in = cv.take<uint64_t>(E > 0 ? E : 1);
    uint64_t* val_out = cv.take<uint64_t>(E > 0 ? E : 1);
    int32_t* rows = row32 ? row32 : row_tmp;
    const int bits = key_bits(N);
    size_t temp_bytes = sort_temp_bytes(E, bits);
    void* temp = cv.take<char>(temp_bytes);
    if (!cv.ok()) return fail(PGLAMD_E_WORKSPACE, "csr_build: workspace carve overflow");

    if (E > 0) {
        hipLaunchKernelGGL(narrow_keys_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, u, u_stride, v, v_stride, E, N, key_in, val_in, range_flag);
        PGLAMD_LAUNCH_CHECK();
        PGLAMD_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, key_in, rows, val_in, val_out, (size_t)E, 0u, (unsigned)bits, st));
    }
    hipLaunchKernelGGL(row_bounds_kernel<int32_t>, dim3(grid_for(E + 1)), dim3(kBlock), 0, st, rows, E, N, indptr);
    PGLAMD_LAUNCH_CHECK();
    if (N > 0) {
        hipLaunchKernelGGL(degree_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, indptr, N, degree);
        PGLAMD_LAUNCH_CHECK();
    }
    if (E > 0) {
        hipLaunchKernelGGL(finish_csr_kernel, dim3(grid_for(E)), dim3(kBlock), 0, st, rows, val_out, E,
                           sorted_v, sorted_u, sorted_eid, col32, eid32);
        PGLAMD_LAUNCH_CHECK();
    }
    return PGLAMD_OK;
}

extern "C" size_t pglamd_unique_segment_workspace_bytes(int64_t num_edges, int64_t num_nodes) {
    (void)num_edges;
    const int64_t N = num_nodes > 0 ? num_nodes : 1;
    return align_up((size_t)N * 8, 256) + align_up(scan_temp_bytes(N), 256) + 512;
}

extern "C" int32_t pglamd_unique_segment(const int64_t* degree, const int64_t* sorted_u, int64_t num_edges,
                                         int64_t num_nodes, int64_t* uniq_ind, int64_t* segment_ids,
                                         int64_t* num_uniq, void* workspace, size_t workspace_bytes, void* stream) {
    if (!degree || !uniq_ind || !num_uniq || (num_edges > 0 && (!sorted_u || !segment_ids)))
        return fail(PGLAMD_E_ARG, "unique_segment: NULL pointer");
    if (!workspace || workspace_bytes < pglamd_unique_segment_workspace_bytes(num_edges, num_nodes))
        return fail(PGLAMD_E_WORKSPACE, "unique_segment: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t N = num_nodes;
    if (N == 0) { PGLAMD_HIP_CHECK(hipMemsetAsync(num_uniq, 0, 8, st)); return PGLAMD_OK; }
    Carver cv(workspace, workspace_bytes);
    int64_t* rank = cv.take<int64_t>(N);
    size_t temp_bytes = scan_temp_bytes(N);
    void* temp = cv.take<char>(temp_bytes);
    auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), NonEmpty{degree});
    PGLAMD_HIP_CHECK(rocprim::exclusive_scan(temp, temp_bytes, it, rank, int64_t(0), (size_t)N, rocprim::plus<int64_t>(), st));
    hipLaunchKernelGGL(uniq_scatter_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, degree, rank, N, uniq_ind, num_uniq);
    PGLAMD_LAUNCH_CHECK();
    if (num_edges > 0) {
        hipLaunchKernelGGL(seg_ids_kernel, dim3(grid_for(num_edges)), dim3(kBlock), 0, st, sorted_u, rank, num_edges, segment_ids);
        PGLAMD_LAUNCH_CHECK();
    }
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_narrow_i64(const int64_t* in, int64_t in_stride, int64_t n, int32_t* out, void* stream) {
    if (n < 0 || (n > 0 && (!in || !out))) return fail(PGLAMD_E_ARG, "narrow_i64: bad argument");
    if (n == 0) return PGLAMD_OK;
    hipLaunchKernelGGL(narrow_i64_kernel, dim3(grid_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), in, in_stride, n, out);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_seg_ptr_from_ids(const void* ids, int32_t ids_i64, int64_t num_rows, int64_t n_seg,
                                           int64_t* seg_ptr, void* stream) {
    if (!seg_ptr || (num_rows > 0 && !ids) || num_rows < 0 || n_seg < 0) return fail(PGLAMD_E_ARG, "seg_ptr_from_ids: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ids_i64)
        hipLaunchKernelGGL(row_bounds_kernel<int64_t>, dim3(grid_for(num_rows + 1)), dim3(kBlock), 0, st,
                           static_cast<const int64_t*>(ids), num_rows, n_seg, seg_ptr);
    else
        hipLaunchKernelGGL(row_bounds_kernel<int32_t>, dim3(grid_for(num_rows + 1)), dim3(kBlock), 0, st,
                           static_cast<const int32_t*>(ids), num_rows, n_seg, seg_ptr);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}
