// host_ops.cpp -- CPU-side helpers of libpglamd (HOST pointers): the host CSR build, id relabel and the halo plan of a
// row-partitioned graph.  (The engine's own partitioner -- what stands behind pgl.partition.metis_partition -- lives in
// partition.cpp; no METIS code is linked, opened or built by the product.)
//
// pglamd_build_index_host <- graph_kernel.build_index (pgl/graph_kernel.pyx:59-88)
// pglamd_map_ids          <- graph_kernel.map_edges / map_nodes (pgl/graph_kernel.pyx:104-138)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>
#include <mutex>
#include <string>

#include "../../include/pgl_amd.h"

namespace pglamd {
int32_t fail(int32_t code, const char* fmt, ...);
}

extern "C" int32_t pglamd_build_index_host(const int64_t* u, int64_t u_stride, const int64_t* v, int64_t v_stride,
                                           int64_t num_edges, int64_t num_nodes, int64_t* degree, int64_t* sorted_v,
                                           int64_t* sorted_u, int64_t* sorted_eid, int64_t* indptr) {
    if (num_edges < 0 || num_nodes < 0 || !indptr || (num_nodes > 0 && !degree) ||
        (num_edges > 0 && (!u || !v || !sorted_v || !sorted_u || !sorted_eid)))
        return pglamd::fail(PGLAMD_E_ARG, "build_index_host: bad argument");
    // histogram -> offsets -> stable placement (a cursor per row walks forward, so equal keys keep
    // their original edge order: the order the reference's counting sort produces)
    std::fill(degree, degree + num_nodes, int64_t(0));
    for (int64_t e = 0; e < num_edges; ++e) {
        const int64_t k = u[e * u_stride];
        if (k < 0 || k >= num_nodes) return pglamd::fail(PGLAMD_E_RANGE, "build_index_host: key %lld out of [0,%lld)", (long long)k, (long long)num_nodes);
        ++degree[k];
    }
    indptr[0] = 0;
    std::partial_sum(degree, degree + num_nodes, indptr + 1);
    std::vector<int64_t> cursor(indptr, indptr + num_nodes);
    for (int64_t e = 0; e < num_edges; ++e) {
        const int64_t k = u[e * u_stride];
        const int64_t slot = cursor[k]++;
        sorted_u[slot] = k;
        sorted_v[slot] = v[e * v_stride];
        sorted_eid[slot] = e;
    }
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_map_ids(const int64_t* keys, const int64_t* vals, int64_t n_keys, const int64_t* in,
                                  int64_t n_in, int64_t* out) {
    if (n_keys < 0 || n_in < 0 || (n_keys > 0 && (!keys || !vals)) || (n_in > 0 && (!in || !out)))
        return pglamd::fail(PGLAMD_E_ARG, "map_ids: bad argument");
    std::unordered_map<int64_t, int64_t> m;
    m.reserve((size_t)n_keys * 2);
    for (int64_t i = 0; i < n_keys; ++i) m[keys[i]] = vals[i];
    for (int64_t i = 0; i < n_in; ++i) {
        auto it = m.find(in[i]);
        out[i] = it == m.end() ? 0 : it->second;   // reference: unordered_map::operator[] -> 0
    }
    return PGLAMD_OK;
}

// ------------------------------------------------------------------------------------------------
// Halo plan of one rank of a row-partitioned graph (host side; SURVEY 8b pglhip_halo_plan_build).  Same arrays, element for
// element, as pgl_amd.distributed.HaloPlan builds with torch (pull plan): relabel so every part owns a contiguous id range
// (apps/GNNAutoScale/graph_partition.py:70-101), keep the in-edges of the owned rows split into local-source and
// halo-source edges, list the distinct halo sources grouped by owner and the owned rows every peer pulls.
// ------------------------------------------------------------------------------------------------
namespace {
struct PlanScratch {
    std::vector<int64_t> offsets, new_id, own_global, loc_rows, loc_cols, loc_eid, hal_src, hal_rows, hal_eid, halo_global,
        send_keys, in_degree, out_degree;
};

int32_t build_plan(const int64_t* src, int64_t ss, const int64_t* dst, int64_t ds, int64_t E, int64_t N, const int64_t* part,
                   int32_t rank, int32_t world, PlanScratch& s) {
    if (E < 0 || N < 0 || world < 1 || rank < 0 || rank >= world || (N > 0 && !part) || (E > 0 && (!src || !dst)))
        return pglamd::fail(PGLAMD_E_ARG, "halo_plan: bad argument");
    s.offsets.assign(world + 1, 0);
    for (int64_t v = 0; v < N; ++v) {
        if (part[v] < 0 || part[v] >= world) return pglamd::fail(PGLAMD_E_RANGE, "halo_plan: part[%lld] = %lld outside [0,%d)", (long long)v, (long long)part[v], world);
        ++s.offsets[part[v] + 1];
    }
    for (int p = 0; p < world; ++p) s.offsets[p + 1] += s.offsets[p];
    std::vector<int64_t> pos(s.offsets.begin(), s.offsets.end() - 1);
    s.new_id.resize(N);
    const int64_t lo = s.offsets[rank], n_own = s.offsets[rank + 1] - lo;
    s.own_global.clear(); s.own_global.reserve(n_own);
    for (int64_t v = 0; v < N; ++v) {                     // stable: ascending original id inside a part
        s.new_id[v] = pos[part[v]]++;
        if (part[v] == rank) s.own_global.push_back(v);
    }
    s.in_degree.assign(n_own, 0); s.out_degree.assign(n_own, 0);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t u = src[e * ss], v = dst[e * ds];
        if (u < 0 || u >= N || v < 0 || v >= N) return pglamd::fail(PGLAMD_E_RANGE, "halo_plan: edge %lld outside [0,%lld)", (long long)e, (long long)N);
        const int64_t pu = part[u], pv = part[v], nu = s.new_id[u], nv = s.new_id[v];
        if (pv == rank) {
            ++s.in_degree[nv - lo];
            if (pu == rank) { s.loc_rows.push_back(nv - lo); s.loc_cols.push_back(nu - lo); s.loc_eid.push_back(e); }
            else { s.hal_src.push_back(nu); s.hal_rows.push_back(nv - lo); s.hal_eid.push_back(e); }
        }
        if (pu == rank) {
            ++s.out_degree[nu - lo];
            if (pv != rank) s.send_keys.push_back(pv * N + nu);
        }
    }
    s.halo_global = s.hal_src;
    std::sort(s.halo_global.begin(), s.halo_global.end());
    s.halo_global.erase(std::unique(s.halo_global.begin(), s.halo_global.end()), s.halo_global.end());
    std::sort(s.send_keys.begin(), s.send_keys.end());
    s.send_keys.erase(std::unique(s.send_keys.begin(), s.send_keys.end()), s.send_keys.end());
    return PGLAMD_OK;
}
}  // namespace

extern "C" int32_t pglamd_halo_plan_sizes(const int64_t* src, int64_t src_stride, const int64_t* dst, int64_t dst_stride,
                                          int64_t num_edges, int64_t num_nodes, const int64_t* part, int32_t rank, int32_t world,
                                          int64_t* sizes) {
    if (!sizes) return pglamd::fail(PGLAMD_E_ARG, "halo_plan_sizes: NULL sizes");
    PlanScratch s;
    const int32_t rc = build_plan(src, src_stride, dst, dst_stride, num_edges, num_nodes, part, rank, world, s);
    if (rc != PGLAMD_OK) return rc;
    sizes[0] = (int64_t)s.own_global.size(); sizes[1] = (int64_t)s.loc_rows.size(); sizes[2] = (int64_t)s.hal_rows.size();
    sizes[3] = (int64_t)s.halo_global.size(); sizes[4] = (int64_t)s.send_keys.size();
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_halo_plan_fill(const int64_t* src, int64_t src_stride, const int64_t* dst, int64_t dst_stride,
                                         int64_t num_edges, int64_t num_nodes, const int64_t* part, int32_t rank, int32_t world,
                                         int64_t* offsets, int64_t* own_global, int64_t* loc_rows, int64_t* loc_cols,
                                         int64_t* hal_rows, int64_t* hal_cols, int64_t* halo_global, int64_t* send_idx,
                                         int64_t* halo_splits, int64_t* pull_splits, int64_t* in_degree, int64_t* out_degree,
                                         int64_t* edge_global) {
    PlanScratch s;
    const int32_t rc = build_plan(src, src_stride, dst, dst_stride, num_edges, num_nodes, part, rank, world, s);
    if (rc != PGLAMD_OK) return rc;
    if (!offsets || !halo_splits || !pull_splits) return pglamd::fail(PGLAMD_E_ARG, "halo_plan_fill: NULL pointer");
    const int64_t N = num_nodes, lo = s.offsets[rank];
    auto put = [](int64_t* d, const std::vector<int64_t>& v) { if (d && !v.empty()) std::memcpy(d, v.data(), v.size() * sizeof(int64_t)); };
    put(offsets, s.offsets); put(own_global, s.own_global); put(loc_rows, s.loc_rows); put(loc_cols, s.loc_cols);
    put(hal_rows, s.hal_rows); put(halo_global, s.halo_global); put(in_degree, s.in_degree); put(out_degree, s.out_degree);
    if (hal_cols)
        for (size_t i = 0; i < s.hal_src.size(); ++i)
            hal_cols[i] = std::lower_bound(s.halo_global.begin(), s.halo_global.end(), s.hal_src[i]) - s.halo_global.begin();
    std::fill(halo_splits, halo_splits + world, (int64_t)0);
    for (int64_t g : s.halo_global) ++halo_splits[std::upper_bound(s.offsets.begin(), s.offsets.end(), g) - s.offsets.begin() - 1];
    std::fill(pull_splits, pull_splits + world, (int64_t)0);
    for (size_t i = 0; i < s.send_keys.size(); ++i) {
        ++pull_splits[s.send_keys[i] / N];
        if (send_idx) send_idx[i] = s.send_keys[i] % N - lo;
    }
    if (edge_global) {
        put(edge_global, s.loc_eid);
        if (!s.hal_eid.empty()) std::memcpy(edge_global + s.loc_eid.size(), s.hal_eid.data(), s.hal_eid.size() * sizeof(int64_t));
    }
    return PGLAMD_OK;
}
