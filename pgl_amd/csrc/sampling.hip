// sampling.hip -- "next" row f3: GPU neighbour sampling + subgraph relabel, the step BEFORE the hot
// path in the GraphSAGE mini-batch loop.  Stands in for
//   Graph.sample_predecessor -> graph_kernel.sample_subset(_with_eid)   (pgl/graph.py:644-688,
//                                                                        pgl/graph_kernel.pyx:266-339)
//   paddle.geometric.sample_neighbors / reindex_graph as used by pgl.sampling.NeighborSampler
//                                                                       (pgl/sampling/sage.py:130-155)
//   graph_kernel.map_nodes / map_edges relabelling                      (pgl/graph_kernel.pyx:104-138)
//
// sample: per seed node, all in-neighbours if degree <= k, else k of them uniformly WITHOUT
//   replacement (Floyd's algorithm, one lane per seed: k is small, k^2/2 compares beat any shared
//   structure).  Randomness is a counter-based hash of (seed, node, draw): reproducible, no state.
// reindex: ids of `nodes` first, then every new neighbour id in order of FIRST APPEARANCE (the
//   contract of reindex_graph): an open-addressing table keeps the minimum position of every key
//   (atomicMin: order-independent, hence deterministic), positions that are firsts are scanned
//   into new ids, a second lookup relabels every neighbour.
#include "common.hpp"

#include "scan.hpp"

namespace pglamd {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(kBlock) void sample_count_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ nodes,
                                                              int64_t n, int64_t k, int64_t* __restrict__ count) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t v = nodes[i];
        const int64_t deg = indptr[v + 1] - indptr[v];
        count[i] = (k < 0 || deg <= k) ? deg : k;
    }
}

constexpr int kMaxSample = 64;     // Floyd's set lives in registers / local array

__global__ __launch_bounds__(kBlock) void sample_fill_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ col,
                                                             const int32_t* __restrict__ eid, const int64_t* __restrict__ nodes,
                                                             int64_t n, int64_t k, uint64_t seed, const int64_t* __restrict__ offsets,
                                                             int64_t* __restrict__ out_nbr, int64_t* __restrict__ out_eid) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t v = nodes[i];
        const int64_t b = indptr[v], deg = indptr[v + 1] - b;
        const int64_t o = offsets[i];
        if (k < 0 || deg <= k) {
            for (int64_t j = 0; j < deg; ++j) {
                out_nbr[o + j] = col[b + j];
                if (out_eid) out_eid[o + j] = eid[b + j];
            }
            continue;
        }
        // Floyd: for j = deg-k .. deg-1: t = U[0, j]; insert t unless already chosen, else insert j
        int64_t chosen[kMaxSample];
        int cnt = 0;
        for (int64_t j = deg - k; j < deg; ++j) {
            const uint64_t r = mix64(seed ^ mix64((uint64_t)v * 0x100000001B3ull + (uint64_t)(j - (deg - k))));
            int64_t t = (int64_t)(r % (uint64_t)(j + 1));
            bool dup = false;
            for (int q = 0; q < cnt; ++q) dup |= (chosen[q] == t);
            if (dup) t = j;
            chosen[cnt++] = t;
        }
        for (int q = 0; q < cnt; ++q) {
            out_nbr[o + q] = col[b + chosen[q]];
            if (out_eid) out_eid[o + q] = eid[b + chosen[q]];
        }
    }
}

// ---- reindex --------------------------------------------------------------------------------------
constexpr int64_t kEmptyKey = -1;

__device__ __forceinline__ uint64_t slot_of(int64_t key, uint64_t mask) { return mix64((uint64_t)key) & mask; }

// table: keys[cap] (int64, -1 = empty), minpos[cap] (int64, init INT64_MAX)
__global__ __launch_bounds__(kBlock) void reindex_insert_kernel(const int64_t* __restrict__ nodes, int64_t n,
                                                                const int64_t* __restrict__ nbrs, int64_t m,
                                                                unsigned long long* keys, unsigned long long* minpos, uint64_t mask) {
    const int64_t total = n + m;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < total; p += (int64_t)gridDim.x * kBlock) {
        const int64_t key = p < n ? nodes[p] : nbrs[p - n];
        uint64_t s = slot_of(key, mask);
        while (true) {
            const unsigned long long prev = atomicCAS(&keys[s], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (prev == (unsigned long long)kEmptyKey || prev == (unsigned long long)key) { atomicMin(&minpos[s], (unsigned long long)p); break; }
            s = (s + 1) & mask;
        }
    }
}

__device__ __forceinline__ uint64_t find_slot(const unsigned long long* keys, int64_t key, uint64_t mask) {
    uint64_t s = slot_of(key, mask);
    while (keys[s] != (unsigned long long)key) s = (s + 1) & mask;
    return s;
}

__global__ __launch_bounds__(kBlock) void reindex_first_kernel(const int64_t* __restrict__ nodes, int64_t n,
                                                               const int64_t* __restrict__ nbrs, int64_t m,
                                                               const unsigned long long* __restrict__ keys,
                                                               const unsigned long long* __restrict__ minpos, uint64_t mask,
                                                               int64_t* __restrict__ is_first) {
    const int64_t total = n + m;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < total; p += (int64_t)gridDim.x * kBlock) {
        const int64_t key = p < n ? nodes[p] : nbrs[p - n];
        is_first[p] = minpos[find_slot(keys, key, mask)] == (unsigned long long)p ? 1 : 0;
    }
}

__global__ __launch_bounds__(kBlock) void reindex_assign_kernel(const int64_t* __restrict__ nodes, int64_t n,
                                                                const int64_t* __restrict__ nbrs, int64_t m,
                                                                const unsigned long long* __restrict__ keys,
                                                                const unsigned long long* __restrict__ minpos, uint64_t mask,
                                                                const int64_t* __restrict__ is_first, const int64_t* __restrict__ rank,
                                                                int64_t* __restrict__ out_src, int64_t* __restrict__ out_nodes,
                                                                int64_t* __restrict__ num_out) {
    const int64_t total = n + m;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < total; p += (int64_t)gridDim.x * kBlock) {
        const int64_t key = p < n ? nodes[p] : nbrs[p - n];
        const int64_t id = rank[minpos[find_slot(keys, key, mask)]];
        if (p >= n) out_src[p - n] = id;
        if (is_first[p]) out_nodes[rank[p]] = key;
        if (p == total - 1) *num_out = rank[p] + is_first[p];
    }
}

static unsigned grid_for(int64_t n) {
    int64_t g = ceil_div(n > 0 ? n : 1, kBlock);
    return (unsigned)(g < 256 * 16 ? g : 256 * 16);
}

static uint64_t table_cap(int64_t total) {
    uint64_t c = 64;
    while (c < (uint64_t)total * 2) c <<= 1;
    return c;
}

static size_t scan_bytes(int64_t total) { return exclusive_scan64_temp_bytes(total); }

}  // namespace pglamd

using namespace pglamd;

extern "C" int32_t pglamd_sample_neighbors_count(const int64_t* indptr, const int64_t* nodes, int64_t n, int64_t k,
                                                 int64_t* count, void* stream) {
    if (n < 0 || (n > 0 && (!indptr || !nodes || !count))) return fail(PGLAMD_E_ARG, "sample_neighbors_count: bad argument");
    if (k > kMaxSample) return fail(PGLAMD_E_SHAPE, "sample_neighbors: sample size %lld > %d", (long long)k, kMaxSample);
    if (n == 0) return PGLAMD_OK;
    hipLaunchKernelGGL(sample_count_kernel, dim3(grid_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), indptr, nodes, n, k, count);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_sample_neighbors_fill(const int64_t* indptr, const int32_t* col, const int32_t* eid, const int64_t* nodes,
                                                int64_t n, int64_t k, uint64_t seed, const int64_t* offsets, int64_t* out_neighbors,
                                                int64_t* out_eids, void* stream) {
    if (n < 0 || (n > 0 && (!indptr || !col || !nodes || !offsets || !out_neighbors)) || (out_eids && !eid))
        return fail(PGLAMD_E_ARG, "sample_neighbors_fill: bad argument");
    if (k > kMaxSample) return fail(PGLAMD_E_SHAPE, "sample_neighbors: sample size %lld > %d", (long long)k, kMaxSample);
    if (n == 0) return PGLAMD_OK;
    hipLaunchKernelGGL(sample_fill_kernel, dim3(grid_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), indptr, col, eid, nodes,
                       n, k, seed, offsets, out_neighbors, out_eids);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

extern "C" size_t pglamd_reindex_workspace_bytes(int64_t num_nodes, int64_t num_neighbors) {
    const int64_t total = num_nodes + num_neighbors;
    const uint64_t cap = table_cap(total);
    return 2 * align_up(cap * 8, 256) + 2 * align_up((size_t)(total > 0 ? total : 1) * 8, 256) + align_up(scan_bytes(total), 256) + 256;
}

extern "C" int32_t pglamd_reindex(const int64_t* nodes, int64_t num_nodes, const int64_t* neighbors, int64_t num_neighbors,
                                  int64_t* reindex_src, int64_t* out_nodes, int64_t* num_out, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    const int64_t total = num_nodes + num_neighbors;
    if (num_nodes < 0 || num_neighbors < 0 || !num_out || (total > 0 && !out_nodes) || (num_nodes > 0 && !nodes) ||
        (num_neighbors > 0 && (!neighbors || !reindex_src)))
        return fail(PGLAMD_E_ARG, "reindex: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (total == 0) { PGLAMD_HIP_CHECK(hipMemsetAsync(num_out, 0, 8, st)); return PGLAMD_OK; }
    if (!workspace || workspace_bytes < pglamd_reindex_workspace_bytes(num_nodes, num_neighbors))
        return fail(PGLAMD_E_WORKSPACE, "reindex: workspace too small");
    const uint64_t cap = table_cap(total);
    Carver cv(workspace, workspace_bytes);
    unsigned long long* keys = cv.take<unsigned long long>(cap);
    unsigned long long* minpos = cv.take<unsigned long long>(cap);
    int64_t* is_first = cv.take<int64_t>(total);
    int64_t* rank = cv.take<int64_t>(total);
    size_t tb = scan_bytes(total);
    void* temp = cv.take<char>(tb);
    PGLAMD_HIP_CHECK(hipMemsetAsync(keys, 0xFF, cap * 8, st));       // all bits set == kEmptyKey == UINT64_MAX (minpos init too)
    PGLAMD_HIP_CHECK(hipMemsetAsync(minpos, 0xFF, cap * 8, st));
    hipLaunchKernelGGL(reindex_insert_kernel, dim3(grid_for(total)), dim3(kBlock), 0, st, nodes, num_nodes, neighbors, num_neighbors, keys, minpos, cap - 1);
    PGLAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(reindex_first_kernel, dim3(grid_for(total)), dim3(kBlock), 0, st, nodes, num_nodes, neighbors, num_neighbors, keys, minpos, cap - 1, is_first);
    PGLAMD_LAUNCH_CHECK();
    { const int32_t rc = exclusive_scan64(LoadI64{is_first}, total, rank, temp, st); if (rc != PGLAMD_OK) return rc; }
    hipLaunchKernelGGL(reindex_assign_kernel, dim3(grid_for(total)), dim3(kBlock), 0, st, nodes, num_nodes, neighbors, num_neighbors, keys, minpos,
                       cap - 1, is_first, rank, reindex_src, out_nodes, num_out);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}
