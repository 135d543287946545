// host_ops.cpp -- CPU-side helpers of libpglamd (HOST pointers): id relabel and the engine's own
// multilevel k-way graph partitioner.
//
// pglamd_map_ids      <- graph_kernel.map_edges / map_nodes (pgl/graph_kernel.pyx:104-138)
// pglamd_partition_kway <- METIS_PartGraphKway as driven by pgl.partition.metis_partition
//                        (pgl/partition.py:37-91 -> pgl/graph_kernel.pyx:434-472).
//
// The reference vendors METIS 5.1 (pgl/third_party/metis, third-party C).  It is NOT copied or
// linked here; the partitioner below is a from-scratch implementation of the same published
// scheme (Karypis & Kumar multilevel k-way: heavy-edge matching coarsening, greedy graph-growing
// initial partition, greedy boundary k-way refinement with a balance constraint).  Its output is
// therefore not bit-identical to METIS; the test-suite compares balance and edge cut against the real
// METIS (built from the reference tree as test infrastructure only).  Partitioning is one-off host setup.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>
#include <mutex>
#include <string>

#include <dlfcn.h>

#include "../../include/pgl_amd.h"

namespace pglamd {
int32_t fail(int32_t code, const char* fmt, ...);
}

namespace {

struct Rng {   // splitmix64 / xorshift: deterministic for a given seed
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    uint64_t below(uint64_t n) { return next() % n; }
};

struct G {
    int64_t n = 0;
    std::vector<int64_t> xadj;     // n+1
    std::vector<int32_t> adj;      // neighbours
    std::vector<int64_t> ew;       // edge weights
    std::vector<int64_t> vw;       // vertex weights
    int64_t tvw = 0;
};

void shuffle_perm(std::vector<int32_t>& p, Rng& rng) {
    for (int64_t i = (int64_t)p.size() - 1; i > 0; --i) std::swap(p[i], p[rng.below((uint64_t)i + 1)]);
}

// heavy-edge matching; returns coarse graph + cmap (fine -> coarse)
bool coarsen(const G& g, int64_t maxvw, Rng& rng, G& c, std::vector<int32_t>& cmap) {
    const int64_t n = g.n;
    std::vector<int32_t> match(n, -1), perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    shuffle_perm(perm, rng);
    // visit low-degree vertices first (helps power-law graphs): bucket by degree class
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) {
        const int64_t da = g.xadj[a + 1] - g.xadj[a], db = g.xadj[b + 1] - g.xadj[b];
        auto cls = [](int64_t d) { int c = 0; while (d > 1) { d >>= 1; ++c; } return c; };
        return cls(da) < cls(db);
    });
    int64_t nc = 0;
    cmap.assign(n, -1);
    for (int64_t ii = 0; ii < n; ++ii) {
        const int32_t v = perm[ii];
        if (match[v] >= 0) continue;
        int32_t best = -1; int64_t bw = -1;
        for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
            const int32_t u = g.adj[p];
            if (u == v || match[u] >= 0) continue;
            if (g.vw[v] + g.vw[u] > maxvw) continue;
            if (g.ew[p] > bw) { bw = g.ew[p]; best = u; }
        }
        if (best < 0) { match[v] = v; cmap[v] = (int32_t)nc++; }
        else { match[v] = best; match[best] = v; cmap[v] = cmap[best] = (int32_t)nc++; }
    }
    if (nc > n * 0.95) return false;   // stalled
    c.n = nc; c.xadj.assign(nc + 1, 0); c.vw.assign(nc, 0); c.tvw = g.tvw;
    c.adj.clear(); c.ew.clear();
    c.adj.reserve(g.adj.size()); c.ew.reserve(g.adj.size());
    std::vector<int64_t> slot(nc, -1);
    std::vector<int32_t> first(nc, -1);
    for (int64_t v = 0; v < n; ++v) if (first[cmap[v]] < 0) first[cmap[v]] = (int32_t)v;
    for (int64_t cv = 0; cv < nc; ++cv) {
        const int32_t v = first[cv];
        const int32_t m = match[v];
        const int64_t start = (int64_t)c.adj.size();
        const int32_t members[2] = {v, m};
        const int cntm = (m == v) ? 1 : 2;
        for (int k = 0; k < cntm; ++k) {
            const int32_t f = members[k];
            c.vw[cv] += g.vw[f];
            for (int64_t p = g.xadj[f]; p < g.xadj[f + 1]; ++p) {
                const int32_t cu = cmap[g.adj[p]];
                if (cu == cv) continue;
                if (slot[cu] < 0) { slot[cu] = (int64_t)c.adj.size(); c.adj.push_back(cu); c.ew.push_back(g.ew[p]); }
                else c.ew[slot[cu]] += g.ew[p];
            }
        }
        for (int64_t p = start; p < (int64_t)c.adj.size(); ++p) slot[c.adj[p]] = -1;
        c.xadj[cv + 1] = (int64_t)c.adj.size();
    }
    return true;
}

int64_t edge_cut(const G& g, const std::vector<int32_t>& part) {
    int64_t cut = 0;
    for (int64_t v = 0; v < g.n; ++v)
        for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p)
            if (part[g.adj[p]] != part[v]) cut += g.ew[p];
    return cut / 2;
}

// greedy graph growing: parts are grown one after the other from a random seed, always absorbing
// the frontier vertex most strongly connected to the growing part.
void initial_partition(const G& g, int k, Rng& rng, std::vector<int32_t>& part) {
    const int64_t n = g.n;
    part.assign(n, -1);
    std::vector<int64_t> conn(n, 0);
    int64_t assigned_w = 0;
    std::vector<int32_t> frontier;
    for (int p = 0; p < k - 1; ++p) {
        const int64_t target = (g.tvw - assigned_w) / (k - p);
        int64_t w = 0;
        frontier.clear();
        // seed: random unassigned vertex
        int32_t seed = -1;
        for (int tries = 0; tries < 64 && seed < 0; ++tries) { int32_t v = (int32_t)rng.below((uint64_t)n); if (part[v] < 0) seed = v; }
        if (seed < 0) for (int64_t v = 0; v < n; ++v) if (part[v] < 0) { seed = (int32_t)v; break; }
        if (seed < 0) break;
        auto absorb = [&](int32_t v) {
            part[v] = p; w += g.vw[v];
            for (int64_t q = g.xadj[v]; q < g.xadj[v + 1]; ++q) {
                const int32_t u = g.adj[q];
                if (part[u] >= 0) continue;
                if (conn[u] == 0) frontier.push_back(u);
                conn[u] += g.ew[q];
            }
        };
        absorb(seed);
        while (w < target) {
            // pick the frontier vertex with max connection (linear scan: coarsest graph is small)
            int64_t bi = -1, bc = -1;
            for (int64_t i = 0; i < (int64_t)frontier.size(); ++i) {
                const int32_t u = frontier[i];
                if (part[u] >= 0) { frontier[i] = frontier.back(); frontier.pop_back(); --i; continue; }
                if (conn[u] > bc) { bc = conn[u]; bi = i; }
            }
            int32_t nxt = -1;
            if (bi >= 0) { nxt = frontier[bi]; frontier[bi] = frontier.back(); frontier.pop_back(); }
            else { for (int64_t v = 0; v < n; ++v) if (part[v] < 0) { nxt = (int32_t)v; break; } }
            if (nxt < 0) break;
            if (w + g.vw[nxt] > target && w > target * 0.9) break;
            absorb(nxt);
        }
        for (int32_t u : frontier) conn[u] = 0;
        for (int64_t v = 0; v < n; ++v) if (part[v] < 0) conn[v] = 0;
        assigned_w += w;
    }
    for (int64_t v = 0; v < n; ++v) if (part[v] < 0) part[v] = k - 1;
}

// greedy k-way boundary refinement with balance constraint
void refine(const G& g, int k, int64_t maxpw, int passes, Rng& rng, std::vector<int32_t>& part) {
    const int64_t n = g.n;
    std::vector<int64_t> pw(k, 0);
    for (int64_t v = 0; v < n; ++v) pw[part[v]] += g.vw[v];
    std::vector<int64_t> conn(k, 0);
    std::vector<int32_t> touched;
    std::vector<int32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    for (int pass = 0; pass < passes; ++pass) {
        shuffle_perm(perm, rng);
        int64_t moves = 0;
        for (int64_t ii = 0; ii < n; ++ii) {
            const int32_t v = perm[ii];
            const int32_t own = part[v];
            touched.clear();
            bool boundary = false;
            for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
                const int32_t q = part[g.adj[p]];
                if (conn[q] == 0) touched.push_back(q);
                conn[q] += g.ew[p];
                if (q != own) boundary = true;
            }
            const bool over = pw[own] > maxpw;
            if (boundary || over) {
                const int64_t cown = conn[own];
                int32_t best = -1; int64_t bgain = over ? INT64_MIN : 0; int64_t bpw = 0;
                for (int32_t q : touched) {
                    if (q == own) continue;
                    if (pw[q] + g.vw[v] > maxpw) continue;
                    const int64_t gain = conn[q] - cown;
                    if (gain > bgain || (gain == bgain && best >= 0 && pw[q] < bpw) ||
                        (gain == 0 && best < 0 && !over && pw[q] + g.vw[v] < pw[own])) {
                        best = q; bgain = gain; bpw = pw[q];
                    }
                }
                if (best < 0 && over) {   // overweight interior vertex: push to the lightest part
                    int32_t lq = (int32_t)(std::min_element(pw.begin(), pw.end()) - pw.begin());
                    if (lq != own && pw[lq] + g.vw[v] <= maxpw) best = lq;
                }
                if (best >= 0 && (bgain > 0 || over || pw[best] + g.vw[v] < pw[own])) {
                    part[v] = best; pw[own] -= g.vw[v]; pw[best] += g.vw[v]; ++moves;
                }
            }
            for (int32_t q : touched) conn[q] = 0;
        }
        if (moves == 0) break;
    }
}

}  // namespace

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

extern "C" int32_t pglamd_partition_kway(int64_t num_nodes, const int64_t* xadj, const int64_t* adjncy,
                                         const int64_t* vwgt, const int64_t* adjwgt, int64_t nparts, uint64_t seed,
                                         int64_t* part, int64_t* edgecut) {
    if (num_nodes < 0 || nparts < 1 || (num_nodes > 0 && (!xadj || !part)))
        return pglamd::fail(PGLAMD_E_ARG, "partition_kway: bad argument");
    if (num_nodes >= INT32_MAX) return pglamd::fail(PGLAMD_E_RANGE, "partition_kway: too many nodes");
    if (num_nodes == 0) { if (edgecut) *edgecut = 0; return PGLAMD_OK; }
    const int k = (int)std::min<int64_t>(nparts, num_nodes);
    if (k == 1) { std::fill(part, part + num_nodes, 0); if (edgecut) *edgecut = 0; return PGLAMD_OK; }

    Rng rng(seed + 1);
    std::vector<G> levels(1);
    G& g0 = levels[0];
    g0.n = num_nodes;
    g0.xadj.assign(xadj, xadj + num_nodes + 1);
    const int64_t m = xadj[num_nodes];
    if (m > 0 && !adjncy) return pglamd::fail(PGLAMD_E_ARG, "partition_kway: adjncy NULL");
    g0.adj.resize(m); g0.ew.resize(m);
    for (int64_t i = 0; i < m; ++i) { g0.adj[i] = (int32_t)adjncy[i]; g0.ew[i] = adjwgt ? adjwgt[i] : 1; }
    g0.vw.resize(num_nodes);
    for (int64_t i = 0; i < num_nodes; ++i) { g0.vw[i] = vwgt ? vwgt[i] : 1; g0.tvw += g0.vw[i]; }

    // ---- coarsening
    const int64_t coarsen_to = std::max<int64_t>(40 * k, 400);
    std::vector<std::vector<int32_t>> cmaps;
    while (levels.back().n > coarsen_to && levels.size() < 40) {
        const G& g = levels.back();
        const int64_t maxvw = std::max<int64_t>(1, (int64_t)(1.5 * g.tvw / coarsen_to));
        G c; std::vector<int32_t> cmap;
        if (!coarsen(g, maxvw, rng, c, cmap)) break;
        levels.push_back(std::move(c));
        cmaps.push_back(std::move(cmap));
    }

    // ---- initial partition on the coarsest graph (best of a few trials)
    const double ub = 1.03;
    const G& gc = levels.back();
    std::vector<int32_t> best_part; int64_t best_cut = -1;
    auto max_pw = [&](const G& g) {
        int64_t mx = 0; for (int64_t v = 0; v < g.n; ++v) mx = std::max(mx, g.vw[v]);
        return std::max<int64_t>((int64_t)(ub * g.tvw / k) + 1, (g.tvw + k - 1) / k + mx / 2);
    };
    for (int trial = 0; trial < 8; ++trial) {
        std::vector<int32_t> p;
        initial_partition(gc, k, rng, p);
        refine(gc, k, max_pw(gc), 10, rng, p);
        const int64_t cut = edge_cut(gc, p);
        if (best_cut < 0 || cut < best_cut) { best_cut = cut; best_part = p; }
    }

    // ---- uncoarsen + refine
    std::vector<int32_t> cur = std::move(best_part);
    for (int64_t lv = (int64_t)levels.size() - 2; lv >= 0; --lv) {
        const G& g = levels[lv];
        std::vector<int32_t> fine(g.n);
        const std::vector<int32_t>& cmap = cmaps[lv];
        for (int64_t v = 0; v < g.n; ++v) fine[v] = cur[cmap[v]];
        cur.swap(fine);
        refine(g, k, max_pw(g), lv == 0 ? 6 : 8, rng, cur);
    }
    for (int64_t v = 0; v < num_nodes; ++v) part[v] = cur[v];
    if (edgecut) *edgecut = edge_cut(levels[0], cur);
    return PGLAMD_OK;
}


// ------------------------------------------------------------------------------------------------
// METIS (the reference's vendored library, built by pgl_amd/_build_metis.py into libpglamd_metis.so next to this library)
// ------------------------------------------------------------------------------------------------
namespace {
typedef int (*metis_kway_fn)(int64_t* nvtxs, int64_t* ncon, int64_t* xadj, int64_t* adjncy, int64_t* vwgt, int64_t* vsize,
                             int64_t* adjwgt, int64_t* nparts, float* tpwgts, float* ubvec, int64_t* options,
                             int64_t* edgecut, int64_t* part);
struct MetisLib {
    std::once_flag once;
    metis_kway_fn kway = nullptr;
    std::string why;
};
MetisLib& metis_lib() { static MetisLib m; return m; }

void load_metis() {
    MetisLib& m = metis_lib();
    std::string path;
    if (const char* e = getenv("PGLAMD_METIS_LIB")) path = e;
    else {
        Dl_info info;
        if (dladdr((void*)&load_metis, &info) && info.dli_fname) {
            path = info.dli_fname;
            const size_t k = path.find_last_of('/');
            path = (k == std::string::npos ? std::string(".") : path.substr(0, k)) + "/libpglamd_metis.so";
        }
    }
    void* h = path.empty() ? nullptr : dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        const char* err = dlerror();               // ONE call: dlerror() clears the message it returns
        m.why = "cannot open " + path + (err ? std::string(": ") + err : std::string());
        return;
    }
    m.kway = (metis_kway_fn)dlsym(h, "METIS_PartGraphKway");
    if (!m.kway) m.why = path + " has no METIS_PartGraphKway";
}
}  // namespace

extern "C" int32_t pglamd_metis_available(void) {
    std::call_once(metis_lib().once, load_metis);
    return metis_lib().kway ? 1 : 0;
}

extern "C" int32_t pglamd_partition_metis(int64_t num_nodes, const int64_t* xadj, const int64_t* adjncy, const int64_t* vwgt,
                                          const int64_t* adjwgt, int64_t nparts, int64_t* part, int64_t* edgecut) {
    if (num_nodes < 0 || nparts < 1 || !part || (num_nodes > 0 && (!xadj || (xadj[num_nodes] > 0 && !adjncy))))
        return pglamd::fail(PGLAMD_E_ARG, "partition_metis: bad argument");
    std::call_once(metis_lib().once, load_metis);
    if (!metis_lib().kway) return pglamd::fail(PGLAMD_E_UNAVAILABLE, "partition_metis: %s", metis_lib().why.c_str());
    if (nparts == 1 || num_nodes == 0) {
        std::fill(part, part + num_nodes, (int64_t)0);
        if (edgecut) *edgecut = 0;
        return PGLAMD_OK;
    }
    // exactly the reference's call (pgl/graph_kernel.pyx:468-471): ncon = 1, no vsize / tpwgts / ubvec, DEFAULT options
    int64_t nv = num_nodes, ncon = 1, np_ = nparts, cut = -1;
    const int rc = metis_lib().kway(&nv, &ncon, const_cast<int64_t*>(xadj), const_cast<int64_t*>(adjncy), const_cast<int64_t*>(vwgt),
                                    nullptr, const_cast<int64_t*>(adjwgt), &np_, nullptr, nullptr, nullptr, &cut, part);
    if (rc != 1) return pglamd::fail(PGLAMD_E_ARG, "partition_metis: METIS_PartGraphKway returned %d", rc);   // METIS_OK == 1
    if (edgecut) *edgecut = cut;
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
