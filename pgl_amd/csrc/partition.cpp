// partition.cpp -- the engine's own multilevel k-way graph partitioner (HOST code, multi-threaded, deterministic).
//
// Stands where the reference calls METIS_PartGraphKway (pgl/partition.py:37-91 -> graph_kernel.metis_partition,
// pgl/graph_kernel.pyx:434-472).  Nothing of METIS is copied or linked: this is a from-scratch partitioner built for the
// graphs this engine is benchmarked on -- power-law graphs with 10^5-edge hubs, a third of the vertices isolated (RMAT) --
// on which matching-based coarsening stalls (round 2's heavy-edge matching took 1 M vertices to 470 k in six levels and
// 30 s).  Scheme (the size-constrained label-propagation multilevel of the KaHIP / KaMinPar family, restated):
//
//   coarsen   size-constrained LABEL PROPAGATION clustering: a vertex joins the neighbouring cluster it is most strongly
//             connected to, as long as the cluster stays below a weight cap.  Leaves whose hub's cluster is full are merged
//             with each other when they favour the same cluster (two-hop clustering), isolated vertices are packed together:
//             a power-law graph shrinks 5-20x per level instead of 1.2x.  Clusters are contracted into the next level.
//   initial   greedy graph growing + greedy k-way boundary refinement on the coarsest graph (<= a few thousand vertices),
//             best of several seeded trials.
//   refine    at every level: label-propagation refinement under the balance constraints; levels below 40 k vertices
//             also get the sequential greedy boundary refinement (cheap there, and it escapes what LP cannot).
//   fill      isolated vertices cost no cut wherever they go: they are placed last, to level the parts.
//
// TWO balance constraints (SURVEY 8e iii + VERDICT r2 item 1d): vwgt = aggregation work of a vertex (in-degree + 1), vwgt2 = 1
// per row -- a rank's time is set by its edges, its memory and its row-wise kernels by its rows, and a single-constraint
// partition of RMAT puts three times the average number of rows on the rank that collects the isolated vertices.
//
// Parallel AND deterministic: every label-propagation round is split into sub-rounds by a hash of the vertex id; inside a
// sub-round all threads compute the moves their vertices want from the frozen labels (the expensive part: adjacency scans),
// then the moves are applied in vertex order against the live weights.  The result depends on the seed only, not on the
// thread count or the schedule -- ranks that partition independently (no process group) arrive at the same parts.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <queue>
#include <thread>
#include <vector>

#include "../../include/pgl_amd.h"

namespace pglamd {
int32_t fail(int32_t code, const char* fmt, ...);
}

namespace {

constexpr int NC = 2;     // balance constraints carried through every level

struct Graph {
    int32_t n = 0;
    std::vector<int64_t> xadj;
    std::vector<int32_t> adj;
    std::vector<int32_t> ew;
    std::vector<int64_t> vw[NC];
    int64_t tvw[NC] = {0, 0};
    int64_t m() const { return (int64_t)adj.size(); }
};

inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(mix64(seed) | 1) {}
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    uint64_t below(uint64_t n) { return next() % n; }
};

// ---- threads ------------------------------------------------------------------------------------------------------------------
int pick_threads(int asked) {
    if (asked <= 0) {
        if (const char* e = getenv("PGLAMD_THREADS")) asked = atoi(e);
    }
    if (asked <= 0) asked = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    return std::max(1, std::min(asked, 64));
}

// fn(t, begin, end) over [0, n) cut at `bounds` (size T+1)
template <typename F>
void run_chunks(const std::vector<int64_t>& bounds, F&& fn) {
    const int T = (int)bounds.size() - 1;
    if (T <= 1) { fn(0, bounds[0], bounds[T]); return; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { fn(t, bounds[t], bounds[t + 1]); });
    fn(0, bounds[0], bounds[1]);
    for (auto& x : th) x.join();
}

std::vector<int64_t> even_bounds(int64_t n, int T) {
    std::vector<int64_t> b(T + 1);
    for (int t = 0; t <= T; ++t) b[t] = n * t / T;
    return b;
}

// vertex ranges of (about) equal adjacency volume: hubs must not make one thread's range ten times the others'
std::vector<int64_t> edge_bounds(const std::vector<int64_t>& xadj, int64_t n, int T) {
    std::vector<int64_t> b(T + 1, n);
    b[0] = 0;
    const int64_t tot = xadj[n] + n;
    for (int t = 1; t < T; ++t) {
        const int64_t want = tot * t / T;
        int64_t lo = b[t - 1], hi = n;
        while (lo < hi) {                     // first v with xadj[v] + v >= want
            const int64_t mid = (lo + hi) / 2;
            if (xadj[mid] + mid < want) lo = mid + 1; else hi = mid;
        }
        b[t] = lo;
    }
    return b;
}

struct Clock {
    bool on; double t0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    Clock() : on(getenv("PGLAMD_PART_VERBOSE") != nullptr), t0(now()) {}
    void lap(const char* what, int64_t n, int64_t m) {
        if (!on) return;
        const double t = now();
        fprintf(stderr, "[partition] %-26s n=%-9lld m=%-10lld %.3f s\n", what, (long long)n, (long long)m, t - t0);
        t0 = t;
    }
};

// per-thread dense rating map: weight per label + the labels touched
struct Rating {
    std::vector<int64_t> w;
    std::vector<int32_t> touched;
    void ensure(int64_t n) { if ((int64_t)w.size() < n) w.assign(n, 0); }
    inline void add(int32_t c, int64_t x) { if (w[c] == 0) touched.push_back(c); w[c] += x; }
    inline void clear() { for (int32_t c : touched) w[c] = 0; touched.clear(); }
};

// ---- coarsening: size-constrained label propagation clustering -------------------------------------------------------------
// clus[v] = cluster label (a vertex id) -> returns the number of clusters and the dense relabelling cmap
int32_t cluster(const Graph& g, const int64_t (&cap)[NC], uint64_t seed, int T, std::vector<int32_t>& cmap) {
    const int32_t n = g.n;
    std::vector<int32_t> clus(n), csize(n, 1), want(n, -1);
    std::iota(clus.begin(), clus.end(), 0);
    std::vector<int64_t> cw[NC] = {g.vw[0], g.vw[1]};
    const std::vector<int64_t> bounds = edge_bounds(g.xadj, n, T);
    std::vector<Rating> rat(T);
    for (auto& r : rat) r.ensure(n);
    constexpr int ITER = 3, SUB = 8;
    for (int it = 0; it < ITER; ++it) {
        int64_t moved_it = 0;
        for (int sr = 0; sr < SUB; ++sr) {
            const uint64_t salt = mix64(seed * 131 + (uint64_t)it * 17 + 3);
            run_chunks(bounds, [&](int t, int64_t b, int64_t e) {
                Rating& R = rat[t];
                for (int64_t v = b; v < e; ++v) {
                    want[v] = -1;
                    if ((int)(mix64((uint64_t)v ^ salt) % SUB) != sr) continue;
                    const int64_t p0 = g.xadj[v], p1 = g.xadj[v + 1];
                    if (p0 == p1) continue;
                    const int32_t own = clus[v];
                    for (int64_t p = p0; p < p1; ++p) { const int32_t u = g.adj[p]; if (u != (int32_t)v) R.add(clus[u], g.ew[p]); }
                    const int64_t own_r = csize[own] > 1 ? R.w[own] : 0;      // a singleton gives up nothing by leaving
                    int32_t best = -1; int64_t br = 0, bw = 0;
                    for (int32_t c : R.touched) {
                        if (c == own) continue;
                        const int64_t r = R.w[c];
                        if (r <= own_r || r < br) continue;                     // must beat staying; ties below: lighter, then smaller label
                        if (cw[0][c] + g.vw[0][v] > cap[0] || cw[1][c] + g.vw[1][v] > cap[1]) continue;
                        if (best < 0 || r > br || cw[0][c] < bw || (cw[0][c] == bw && c < best)) { best = c; br = r; bw = cw[0][c]; }
                    }
                    R.clear();
                    want[v] = best;
                }
            });
            // apply in vertex order against the live weights
            for (int32_t v = 0; v < n; ++v) {
                const int32_t c = want[v];
                if (c < 0) continue;
                const int32_t own = clus[v];
                if (cw[0][c] + g.vw[0][v] > cap[0] || cw[1][c] + g.vw[1][v] > cap[1]) continue;
                if (csize[c] == 0) continue;      // emptied earlier in this sub-round (two singletons that chose each other: the first one moved)
                for (int k = 0; k < NC; ++k) { cw[k][c] += g.vw[k][v]; cw[k][own] -= g.vw[k][v]; }
                --csize[own]; ++csize[c];
                clus[v] = c;
                ++moved_it;
            }
        }
        if (moved_it < n / 200) break;
    }
    // two-hop clustering: singletons that favour the same (full) cluster are merged with each other; isolated vertices are packed
    {
        std::vector<int32_t> fav(n, -1);
        run_chunks(bounds, [&](int t, int64_t b, int64_t e) {
            Rating& R = rat[t];
            for (int64_t v = b; v < e; ++v) {
                if (csize[clus[v]] != 1) continue;
                const int64_t p0 = g.xadj[v], p1 = g.xadj[v + 1];
                if (p0 == p1) { fav[v] = -2; continue; }            // isolated
                for (int64_t p = p0; p < p1; ++p) { const int32_t u = g.adj[p]; if (u != (int32_t)v) R.add(clus[u], g.ew[p]); }
                int32_t best = -1; int64_t br = 0;
                for (int32_t c : R.touched) if (R.w[c] > br || (R.w[c] == br && c < best)) { best = c; br = R.w[c]; }
                R.clear();
                fav[v] = best;
            }
        });
        std::vector<int32_t> open(n, -1);         // favourite cluster -> the singleton currently collecting its leaves
        int32_t iso_open = -1;
        for (int32_t v = 0; v < n; ++v) {
            const int32_t f = fav[v];
            if (f == -1 || csize[clus[v]] != 1) continue;
            int32_t& slot = f == -2 ? iso_open : open[f];
            if (slot >= 0 && slot != v && cw[0][clus[slot]] + g.vw[0][v] <= cap[0] && cw[1][clus[slot]] + g.vw[1][v] <= cap[1]) {
                const int32_t c = clus[slot], own = clus[v];
                for (int k = 0; k < NC; ++k) { cw[k][c] += g.vw[k][v]; cw[k][own] -= g.vw[k][v]; }
                --csize[own]; ++csize[c];
                clus[v] = c;
            } else {
                slot = v;
            }
        }
    }
    // dense ids in order of first appearance
    std::vector<int32_t> id(n, -1);
    cmap.resize(n);
    int32_t nc = 0;
    for (int32_t v = 0; v < n; ++v) {
        int32_t& x = id[clus[v]];
        if (x < 0) x = nc++;
        cmap[v] = x;
    }
    return nc;
}

void contract(const Graph& g, const std::vector<int32_t>& cmap, int32_t nc, int T, Graph& c) {
    const int32_t n = g.n;
    c.n = nc;
    for (int k = 0; k < NC; ++k) { c.vw[k].assign(nc, 0); c.tvw[k] = g.tvw[k]; }
    // members of every coarse vertex, grouped (counting sort by coarse id; stable => deterministic)
    std::vector<int64_t> cstart(nc + 1, 0);
    for (int32_t v = 0; v < n; ++v) ++cstart[cmap[v] + 1];
    for (int32_t i = 0; i < nc; ++i) cstart[i + 1] += cstart[i];
    std::vector<int32_t> order(n);
    {
        std::vector<int64_t> cur(cstart.begin(), cstart.end() - 1);
        for (int32_t v = 0; v < n; ++v) order[cur[cmap[v]]++] = v;
    }
    // work per coarse vertex = its members' adjacency: balance the thread ranges on that
    std::vector<int64_t> work(nc + 1, 0);
    for (int32_t v = 0; v < n; ++v) work[cmap[v] + 1] += g.xadj[v + 1] - g.xadj[v];
    for (int32_t i = 0; i < nc; ++i) work[i + 1] += work[i];
    const std::vector<int64_t> bounds = edge_bounds(work, nc, T);
    std::vector<std::vector<int32_t>> ladj(T), lew(T);
    std::vector<int64_t> deg(nc, 0);
    run_chunks(bounds, [&](int t, int64_t b, int64_t e) {
        std::vector<int32_t> slot(nc, -1);
        std::vector<int32_t>& A = ladj[t];
        std::vector<int32_t>& W = lew[t];
        A.reserve((size_t)(work[e] - work[b]) / 2 + 16); W.reserve(A.capacity());
        for (int64_t cv = b; cv < e; ++cv) {
            const int64_t start = (int64_t)A.size();
            for (int64_t q = cstart[cv]; q < cstart[cv + 1]; ++q) {
                const int32_t f = order[q];
                for (int k = 0; k < NC; ++k) c.vw[k][cv] += g.vw[k][f];
                for (int64_t p = g.xadj[f]; p < g.xadj[f + 1]; ++p) {
                    const int32_t cu = cmap[g.adj[p]];
                    if (cu == (int32_t)cv) continue;
                    if (slot[cu] < 0) { slot[cu] = (int32_t)(A.size() - start); A.push_back(cu); W.push_back(g.ew[p]); }
                    else { int32_t& w = W[start + slot[cu]]; w = (int32_t)std::min<int64_t>((int64_t)w + g.ew[p], INT32_MAX); }   // saturating: > 2^31 merged weight between two coarse vertices (graphs beyond ~1e9 edges) must not wrap negative
                }
            }
            deg[cv] = (int64_t)A.size() - start;
            for (int64_t p = start; p < (int64_t)A.size(); ++p) slot[A[p]] = -1;
        }
    });
    c.xadj.assign(nc + 1, 0);
    for (int32_t i = 0; i < nc; ++i) c.xadj[i + 1] = c.xadj[i] + deg[i];
    c.adj.resize(c.xadj[nc]); c.ew.resize(c.xadj[nc]);
    run_chunks(bounds, [&](int t, int64_t b, int64_t e) {
        if (b >= e) return;
        std::copy(ladj[t].begin(), ladj[t].end(), c.adj.begin() + c.xadj[b]);
        std::copy(lew[t].begin(), lew[t].end(), c.ew.begin() + c.xadj[b]);
    });
}

// ---- balance bookkeeping --------------------------------------------------------------------------------------------------
struct Limits {
    int64_t maxpw[NC];
};

Limits limits_for(const Graph& g, int k, const double (&ub)[NC]) {
    Limits L;
    for (int c = 0; c < NC; ++c) {
        int64_t mx = 0;
        for (int32_t v = 0; v < g.n; ++v) mx = std::max(mx, g.vw[c][v]);
        const int64_t even = (g.tvw[c] + k - 1) / k;
        L.maxpw[c] = std::max<int64_t>((int64_t)(ub[c] * (double)g.tvw[c] / k), even + mx / 2);
    }
    return L;
}

int64_t edge_cut(const Graph& g, const std::vector<int32_t>& part, int T) {
    const std::vector<int64_t> bounds = edge_bounds(g.xadj, g.n, T);
    std::vector<int64_t> cuts(T, 0);
    run_chunks(bounds, [&](int t, int64_t b, int64_t e) {
        int64_t cut = 0;
        for (int64_t v = b; v < e; ++v)
            for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p)
                if (part[g.adj[p]] != part[v]) cut += g.ew[p];
        cuts[t] = cut;
    });
    return std::accumulate(cuts.begin(), cuts.end(), (int64_t)0) / 2;
}

// ---- initial partition of the coarsest graph: recursive bisection, each bisection = greedy growing + FM --------------------
// (greedy k-way growing followed by greedy refinement, round 2's scheme, gets stuck when a part has to be made of whole
//  communities: on the k = 2 fixture it ended 2.7x above METIS's cut from every seed.  FM accepts negative-gain moves and
//  rolls back to the best prefix, which is what climbing out of such a state takes.)
struct Bisector {
    const Graph& g;
    double frac0;                  // share of the total weight side 0 should get
    double tol;                    // allowed excess over a side's target (e.g. 0.03)
    int64_t maxw[2][NC];
    std::vector<int8_t> side;
    std::vector<int64_t> gain;
    int64_t pw[2][NC];

    Bisector(const Graph& g_, double f, double tol_) : g(g_), frac0(f), tol(tol_) {
        for (int c = 0; c < NC; ++c) {
            int64_t mx = 0;
            for (int32_t v = 0; v < g.n; ++v) mx = std::max(mx, g.vw[c][v]);
            const double t0 = frac0 * (double)g.tvw[c], t1 = (double)g.tvw[c] - t0;
            maxw[0][c] = std::max<int64_t>((int64_t)(t0 * (1.0 + tol)), (int64_t)t0 + mx / 2 + (g.tvw[c] > 0));
            maxw[1][c] = std::max<int64_t>((int64_t)(t1 * (1.0 + tol)), (int64_t)t1 + mx / 2 + (g.tvw[c] > 0));
        }
    }
    void weigh() {
        for (int s = 0; s < 2; ++s) for (int c = 0; c < NC; ++c) pw[s][c] = 0;
        for (int32_t v = 0; v < g.n; ++v) for (int c = 0; c < NC; ++c) pw[side[v]][c] += g.vw[c][v];
    }
    // how far the heavier side is over its bound, as a fraction (0 = feasible)
    double excess() const {
        double x = 0;
        for (int s = 0; s < 2; ++s) for (int c = 0; c < NC; ++c)
            if (pw[s][c] > maxw[s][c]) x = std::max(x, (double)(pw[s][c] - maxw[s][c]) / std::max<double>(1, (double)maxw[s][c]));
        return x;
    }
    int64_t cut() const {
        int64_t c = 0;
        for (int32_t v = 0; v < g.n; ++v)
            for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) if (side[g.adj[p]] != side[v]) c += g.ew[p];
        return c / 2;
    }
    void grow(Rng& rng) {
        const int32_t n = g.n;
        side.assign(n, 1);
        std::vector<int64_t> conn(n, 0);
        using Item = std::pair<int64_t, int32_t>;
        std::priority_queue<Item> pq;
        int64_t w[NC] = {0, 0};
        const double t0[NC] = {frac0 * (double)g.tvw[0], frac0 * (double)g.tvw[1]};
        auto absorb = [&](int32_t v) {
            side[v] = 0;
            for (int c = 0; c < NC; ++c) w[c] += g.vw[c][v];
            for (int64_t q = g.xadj[v]; q < g.xadj[v + 1]; ++q) {
                const int32_t u = g.adj[q];
                if (side[u] == 0) continue;
                conn[u] += g.ew[q];
                pq.emplace(conn[u], u);
            }
        };
        int32_t scan = 0;
        bool seeded = false;
        while ((double)w[0] < t0[0] && (g.tvw[1] == 0 || (double)w[1] < t0[1] * 1.05 + 1)) {
            int32_t nxt = -1;
            while (!pq.empty()) {
                const Item it = pq.top(); pq.pop();
                if (side[it.second] == 1 && conn[it.second] == it.first) { nxt = it.second; break; }
            }
            if (nxt < 0) {
                if (!seeded) {
                    for (int tries = 0; tries < 64 && nxt < 0; ++tries) { const int32_t v = (int32_t)rng.below((uint64_t)n); if (side[v] == 1 && g.xadj[v + 1] > g.xadj[v]) nxt = v; }
                    seeded = true;
                }
                if (nxt < 0) {                      // (next component)
                    while (scan < n && side[scan] == 0) ++scan;
                    if (scan >= n) break;
                    nxt = scan;
                }
            }
            if ((double)(w[0] + g.vw[0][nxt]) > t0[0] * (1.0 + tol) && w[0] > 0) { if (pq.empty()) break; else continue; }
            absorb(nxt);
        }
    }
    // one FM pass with rollback; returns true when it improved (balance first, then cut)
    bool fm_pass() {
        const int32_t n = g.n;
        gain.assign(n, 0);
        for (int32_t v = 0; v < n; ++v) {
            int64_t gv = 0;
            for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) gv += side[g.adj[p]] != side[v] ? g.ew[p] : -(int64_t)g.ew[p];
            gain[v] = gv;
        }
        using Item = std::pair<int64_t, int32_t>;
        std::priority_queue<Item> pq[2];
        for (int32_t v = 0; v < n; ++v) pq[side[v]].emplace(gain[v], v);
        std::vector<char> locked(n, 0);
        std::vector<int32_t> moved;
        const double ex0 = excess();
        double best_ex = ex0; int64_t cum = 0, best_cum = 0; size_t best_len = 0;
        const size_t patience = (size_t)std::max<int64_t>(64, n / 16);
        auto top = [&](int s) -> int32_t {
            while (!pq[s].empty()) {
                const Item it = pq[s].top();
                if (!locked[it.second] && side[it.second] == s && gain[it.second] == it.first) return it.second;
                pq[s].pop();
            }
            return -1;
        };
        auto fits = [&](int to, int32_t v) {
            for (int c = 0; c < NC; ++c) if (pw[to][c] + g.vw[c][v] > maxw[to][c]) return false;
            return true;
        };
        while (moved.size() - best_len < patience) {
            const int32_t c0 = top(0), c1 = top(1);
            int from = -1;
            const bool over0 = pw[0][0] > maxw[0][0] || pw[0][1] > maxw[0][1], over1 = pw[1][0] > maxw[1][0] || pw[1][1] > maxw[1][1];
            const bool ok0 = c0 >= 0 && (fits(1, c0) || over0) && !over1, ok1 = c1 >= 0 && (fits(0, c1) || over1) && !over0;
            const bool f0 = c0 >= 0 && over0, f1 = c1 >= 0 && over1;                // an overweight side must shed
            if (f0 && !over1) from = 0;
            else if (f1 && !over0) from = 1;
            else if (ok0 && ok1) from = gain[c0] > gain[c1] ? 0 : gain[c1] > gain[c0] ? 1 : (pw[0][0] * (1.0 - frac0) >= pw[1][0] * frac0 ? 0 : 1);
            else if (ok0) from = 0;
            else if (ok1) from = 1;
            else break;
            const int32_t v = from == 0 ? c0 : c1;
            pq[from].pop();
            locked[v] = 1;
            side[v] = (int8_t)(1 - from);
            for (int c = 0; c < NC; ++c) { pw[from][c] -= g.vw[c][v]; pw[1 - from][c] += g.vw[c][v]; }
            cum += gain[v];
            moved.push_back(v);
            for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
                const int32_t u = g.adj[p];
                if (locked[u]) continue;
                gain[u] += side[u] == side[v] ? -2 * (int64_t)g.ew[p] : 2 * (int64_t)g.ew[p];
                pq[side[u]].emplace(gain[u], u);
            }
            const double ex = excess();
            if (ex < best_ex - 1e-12 || (ex <= best_ex + 1e-12 && cum > best_cum)) { best_ex = ex; best_cum = cum; best_len = moved.size(); }
        }
        for (size_t i = moved.size(); i > best_len; --i) {           // roll back to the best prefix
            const int32_t v = moved[i - 1];
            const int s = side[v];
            side[v] = (int8_t)(1 - s);
            for (int c = 0; c < NC; ++c) { pw[s][c] -= g.vw[c][v]; pw[1 - s][c] += g.vw[c][v]; }
        }
        return best_len > 0 && (best_ex < ex0 - 1e-12 || best_cum > 0);
    }
};

void induced(const Graph& g, const std::vector<int8_t>& side, int s, Graph& out, std::vector<int32_t>& ids) {
    std::vector<int32_t> newid(g.n, -1);
    ids.clear();
    for (int32_t v = 0; v < g.n; ++v) if (side[v] == s) { newid[v] = (int32_t)ids.size(); ids.push_back(v); }
    out.n = (int32_t)ids.size();
    out.xadj.assign(out.n + 1, 0);
    out.adj.clear(); out.ew.clear();
    for (int c = 0; c < NC; ++c) { out.vw[c].resize(out.n); out.tvw[c] = 0; }
    for (int32_t i = 0; i < out.n; ++i) {
        const int32_t v = ids[i];
        for (int c = 0; c < NC; ++c) { out.vw[c][i] = g.vw[c][v]; out.tvw[c] += g.vw[c][v]; }
        for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
            const int32_t u = newid[g.adj[p]];
            if (u >= 0) { out.adj.push_back(u); out.ew.push_back(g.ew[p]); }
        }
        out.xadj[i + 1] = (int64_t)out.adj.size();
    }
}

void recursive_bisection(const Graph& g, int k, int base, double tol, uint64_t seed, const std::vector<int32_t>& orig,
                         std::vector<int32_t>& part) {
    if (k == 1 || g.n == 0) { for (int32_t v = 0; v < g.n; ++v) part[orig[v]] = base; return; }
    const int k0 = k / 2, k1 = k - k0;
    const int trials = g.n <= 4000 ? 10 : g.n <= 40000 ? 4 : 2;
    std::vector<int8_t> best_side; int64_t best_cut = -1; double best_ex = 0;
    for (int t = 0; t < trials; ++t) {
        Rng rng(seed * 7919 + (uint64_t)t * 104729 + (uint64_t)k);
        Bisector b(g, (double)k0 / k, tol);
        b.grow(rng);
        b.weigh();
        for (int pass = 0; pass < 10; ++pass) if (!b.fm_pass()) break;
        const int64_t cut = b.cut();
        const double ex = b.excess();
        if (best_cut < 0 || ex < best_ex - 1e-12 || (ex <= best_ex + 1e-12 && cut < best_cut)) { best_cut = cut; best_ex = ex; best_side = b.side; }
    }
    for (int s = 0; s < 2; ++s) {
        Graph sub; std::vector<int32_t> ids;
        induced(g, best_side, s, sub, ids);
        std::vector<int32_t> sub_orig(ids.size());
        for (size_t i = 0; i < ids.size(); ++i) sub_orig[i] = orig[ids[i]];
        recursive_bisection(sub, s == 0 ? k0 : k1, s == 0 ? base : base + k0, tol, seed + 17 * (s + 1), sub_orig, part);
    }
}

// sequential greedy k-way boundary refinement under both constraints (small graphs)
void refine_seq(const Graph& g, int k, const Limits& L, int passes, Rng& rng, std::vector<int32_t>& part) {
    const int32_t n = g.n;
    std::vector<int64_t> pw[NC];
    for (int c = 0; c < NC; ++c) pw[c].assign(k, 0);
    for (int32_t v = 0; v < n; ++v) for (int c = 0; c < NC; ++c) pw[c][part[v]] += g.vw[c][v];
    std::vector<int64_t> conn(k, 0);
    std::vector<int32_t> touched, perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    auto fits = [&](int q, int32_t v) { return pw[0][q] + g.vw[0][v] <= L.maxpw[0] && pw[1][q] + g.vw[1][v] <= L.maxpw[1]; };
    for (int pass = 0; pass < passes; ++pass) {
        for (int64_t i = n - 1; i > 0; --i) std::swap(perm[i], perm[rng.below((uint64_t)i + 1)]);
        int64_t moves = 0;
        for (int32_t ii = 0; ii < n; ++ii) {
            const int32_t v = perm[ii], own = part[v];
            touched.clear();
            bool boundary = false;
            for (int64_t p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
                const int32_t q = part[g.adj[p]];
                if (conn[q] == 0) touched.push_back(q);
                conn[q] += g.ew[p];
                if (q != own) boundary = true;
            }
            const bool over = pw[0][own] > L.maxpw[0] || pw[1][own] > L.maxpw[1];
            if (boundary || over) {
                const int64_t cown = conn[own];
                int32_t best = -1; int64_t bgain = over ? INT64_MIN : 0, bpw = 0;
                for (int32_t q : touched) {
                    if (q == own || !fits(q, v)) continue;
                    const int64_t gain = conn[q] - cown;
                    if (gain > bgain || (gain == bgain && best >= 0 && pw[0][q] < bpw) ||
                        (gain == 0 && best < 0 && !over && pw[0][q] + g.vw[0][v] < pw[0][own])) {
                        best = q; bgain = gain; bpw = pw[0][q];
                    }
                }
                if (best < 0 && over) {              // overweight part, no admissible neighbour part: the lightest part that fits
                    int64_t lw = INT64_MAX;
                    for (int q = 0; q < k; ++q) if (q != own && fits(q, v) && pw[0][q] < lw) { lw = pw[0][q]; best = q; }
                }
                if (best >= 0 && (bgain > 0 || over || pw[0][best] + g.vw[0][v] < pw[0][own])) {
                    part[v] = best; ++moves;
                    for (int c = 0; c < NC; ++c) { pw[c][own] -= g.vw[c][v]; pw[c][best] += g.vw[c][v]; }
                }
            }
            for (int32_t q : touched) conn[q] = 0;
        }
        if (moves == 0) break;
    }
}

// parallel, deterministic label-propagation refinement (large levels)
void refine_lp(const Graph& g, int k, const Limits& L, int passes, uint64_t seed, int T, std::vector<int32_t>& part) {
    const int32_t n = g.n;
    std::vector<int64_t> pw[NC];
    for (int c = 0; c < NC; ++c) pw[c].assign(k, 0);
    for (int32_t v = 0; v < n; ++v) for (int c = 0; c < NC; ++c) pw[c][part[v]] += g.vw[c][v];
    const std::vector<int64_t> bounds = edge_bounds(g.xadj, n, T);
    std::vector<int32_t> want(n, -1);
    std::vector<int64_t> gain(n, 0);
    constexpr int SUB = 8;
    for (int pass = 0; pass < passes; ++pass) {
        int64_t moves = 0;
        const uint64_t salt = mix64(seed * 977 + (uint64_t)pass * 31 + 7);
        for (int sr = 0; sr < SUB; ++sr) {
            run_chunks(bounds, [&](int, int64_t b, int64_t e) {
                std::vector<int64_t> conn(k, 0);
                for (int64_t v = b; v < e; ++v) {
                    want[v] = -1;
                    if ((int)(mix64((uint64_t)v ^ salt) % SUB) != sr) continue;
                    const int64_t p0 = g.xadj[v], p1 = g.xadj[v + 1];
                    if (p0 == p1) continue;
                    const int32_t own = part[v];
                    bool boundary = false;
                    for (int64_t p = p0; p < p1; ++p) { const int32_t q = part[g.adj[p]]; conn[q] += g.ew[p]; boundary |= q != own; }
                    const bool over = pw[0][own] > L.maxpw[0] || pw[1][own] > L.maxpw[1];
                    if (boundary || over) {
                        int32_t best = -1; int64_t bg = over ? INT64_MIN : 0, bpw = 0;
                        for (int q = 0; q < k; ++q) {
                            if (q == own || (conn[q] == 0 && !over)) continue;
                            if (pw[0][q] + g.vw[0][v] > L.maxpw[0] || pw[1][q] + g.vw[1][v] > L.maxpw[1]) continue;
                            const int64_t gq = conn[q] - conn[own];
                            if (gq > bg || (gq == bg && best >= 0 && pw[0][q] < bpw) ||
                                (gq == 0 && best < 0 && !over && pw[0][q] + g.vw[0][v] < pw[0][own])) { best = q; bg = gq; bpw = pw[0][q]; }
                        }
                        if (best >= 0) { want[v] = best; gain[v] = bg; }
                    }
                    if (boundary || over) std::fill(conn.begin(), conn.end(), 0);
                    else conn[own] = 0;
                }
            });
            for (int32_t v = 0; v < n; ++v) {
                const int32_t q = want[v];
                if (q < 0) continue;
                const int32_t own = part[v];
                if (pw[0][q] + g.vw[0][v] > L.maxpw[0] || pw[1][q] + g.vw[1][v] > L.maxpw[1]) continue;
                const bool over = pw[0][own] > L.maxpw[0] || pw[1][own] > L.maxpw[1];
                if (!(gain[v] > 0 || over || pw[0][q] + g.vw[0][v] < pw[0][own])) continue;
                part[v] = q; ++moves;
                for (int c = 0; c < NC; ++c) { pw[c][own] -= g.vw[c][v]; pw[c][q] += g.vw[c][v]; }
            }
        }
        if (getenv("PGLAMD_PART_VERBOSE")) {
            int64_t mx0 = 0, mx1 = 0;
            for (int q = 0; q < k; ++q) { mx0 = std::max(mx0, pw[0][q]); mx1 = std::max(mx1, pw[1][q]); }
            fprintf(stderr, "[partition]     lp pass %d: %lld moves, max pw %lld / %lld (limits %lld / %lld)\n", pass, (long long)moves,
                    (long long)mx0, (long long)mx1, (long long)L.maxpw[0], (long long)L.maxpw[1]);
        }
        if (moves < std::max<int64_t>(1, n / 2000)) break;
    }
}

// isolated vertices last: wherever they go costs no cut, so they level the parts (largest first)
void fill_isolated(const Graph& g, int k, const double (&ub)[NC], std::vector<int32_t>& part) {
    std::vector<int32_t> iso;
    for (int32_t v = 0; v < g.n; ++v) if (g.xadj[v] == g.xadj[v + 1]) iso.push_back(v);
    if (iso.empty()) return;
    std::vector<int64_t> pw[NC];
    for (int c = 0; c < NC; ++c) pw[c].assign(k, 0);
    std::vector<char> is_iso(g.n, 0);
    for (int32_t v : iso) is_iso[v] = 1;
    for (int32_t v = 0; v < g.n; ++v) if (!is_iso[v]) for (int c = 0; c < NC; ++c) pw[c][part[v]] += g.vw[c][v];
    std::stable_sort(iso.begin(), iso.end(), [&](int32_t a, int32_t b) { return g.vw[0][a] + g.vw[1][a] > g.vw[0][b] + g.vw[1][b]; });
    const double avg[NC] = {std::max(1.0, (double)g.tvw[0] / k), std::max(1.0, (double)g.tvw[1] / k)};
    for (int32_t v : iso) {
        // the part with the fewest rows among those that stay within the FIRST constraint's bound; none: the lightest part
        int best = -1; double bl = 1e300;
        for (int q = 0; q < k; ++q) {
            if ((pw[0][q] + g.vw[0][v]) / avg[0] > ub[0]) continue;
            const double load = (pw[1][q] + g.vw[1][v]) / avg[1] + 1e-9 * (pw[0][q] / avg[0]);
            if (load < bl) { bl = load; best = q; }
        }
        if (best < 0) {
            for (int q = 0; q < k; ++q) { const double load = (pw[0][q] + g.vw[0][v]) / avg[0]; if (load < bl) { bl = load; best = q; } }
        }
        part[v] = best;
        for (int c = 0; c < NC; ++c) pw[c][best] += g.vw[c][v];
    }
}

int32_t partition_core(Graph& g0, int k, const double (&ub)[NC], uint64_t seed, int threads, int64_t* part_out, int64_t* edgecut) {
    const int T = pick_threads(threads);
    Clock clk;
    clk.lap("input", g0.n, g0.m());
    // isolated vertices are placed LAST (fill_isolated): during the multilevel phase they weigh nothing, otherwise the packs
    // they are clustered into fill whole parts' row budgets and block every refinement move (RMAT: a third of the rows)
    std::vector<int32_t> iso;
    std::vector<int64_t> iso_w[NC];
    for (int32_t v = 0; v < g0.n; ++v)
        if (g0.xadj[v] == g0.xadj[v + 1]) {
            iso.push_back(v);
            for (int c = 0; c < NC; ++c) { iso_w[c].push_back(g0.vw[c][v]); g0.tvw[c] -= g0.vw[c][v]; g0.vw[c][v] = 0; }
        }
    std::vector<Graph> levels;
    levels.push_back(std::move(g0));
    std::vector<std::vector<int32_t>> cmaps;
    int64_t coarsen_to = std::max<int64_t>(25 * (int64_t)k, 200);     // (METIS: max(n / (20 log2 k), 30 k))
    if (const char* e = getenv("PGLAMD_PART_COARSEN_TO")) coarsen_to = atoll(e);
    while (levels.back().n > coarsen_to && levels.size() < 30) {
        const Graph& g = levels.back();
        int64_t cap[NC];
        for (int c = 0; c < NC; ++c) {
            int64_t mx = 0;
            for (int32_t v = 0; v < g.n; ++v) mx = std::max(mx, g.vw[c][v]);
            cap[c] = std::max<int64_t>(mx, (int64_t)(g.tvw[c] / (double)coarsen_to) + 1);
        }
        std::vector<int32_t> cmap;
        const int32_t nc = cluster(g, cap, seed + levels.size() * 7919, T, cmap);
        if (nc > g.n * 0.95) break;                 // stalled
        Graph c;
        contract(g, cmap, nc, T, c);
        levels.push_back(std::move(c));
        cmaps.push_back(std::move(cmap));
        clk.lap("coarsen", levels.back().n, levels.back().m());
    }
    const Graph& gc = levels.back();
    std::vector<int32_t> cur;
    {
        const Limits L = limits_for(gc, k, ub);
        int levels_rb = 1;
        while ((1 << levels_rb) < k) ++levels_rb;
        const double tol = (ub[0] - 1.0) / levels_rb;           // per bisection, so that the k parts end within ub
        std::vector<int32_t> orig(gc.n);
        std::iota(orig.begin(), orig.end(), 0);
        cur.assign(gc.n, 0);
        recursive_bisection(gc, k, 0, tol, seed + 1, orig, cur);
        if (clk.on) fprintf(stderr, "[partition]   recursive bisection cut %lld\n", (long long)edge_cut(gc, cur, 1));
        Rng rng(seed * 1000003 + 1);
        refine_seq(gc, k, L, 12, rng, cur);
        if (clk.on) fprintf(stderr, "[partition]   + k-way refinement cut %lld\n", (long long)edge_cut(gc, cur, 1));
        clk.lap("initial partition", gc.n, gc.m());
    }
    for (int64_t lv = (int64_t)levels.size() - 2; lv >= 0; --lv) {
        const Graph& g = levels[lv];
        std::vector<int32_t> fine(g.n);
        const std::vector<int32_t>& cmap = cmaps[lv];
        for (int32_t v = 0; v < g.n; ++v) fine[v] = cur[cmap[v]];
        cur.swap(fine);
        const Limits L = limits_for(g, k, ub);
        static const int lp_passes = getenv("PGLAMD_PART_LP_PASSES") ? atoi(getenv("PGLAMD_PART_LP_PASSES")) : 6;
        static const int64_t seq_n = getenv("PGLAMD_PART_SEQ_N") ? atoll(getenv("PGLAMD_PART_SEQ_N")) : 40000;
        refine_lp(g, k, L, lp_passes, seed + lv * 131, T, cur);
        if (g.n <= seq_n) { Rng rng(seed * 31 + lv + 5); refine_seq(g, k, L, 6, rng, cur); }
        if (clk.on) fprintf(stderr, "[partition]   level %lld cut %lld\n", (long long)lv, (long long)edge_cut(g, cur, T));
        clk.lap("refine", g.n, g.m());
    }
    for (size_t i = 0; i < iso.size(); ++i)
        for (int c = 0; c < NC; ++c) { levels[0].vw[c][iso[i]] = iso_w[c][i]; levels[0].tvw[c] += iso_w[c][i]; }
    fill_isolated(levels[0], k, ub, cur);
    for (int32_t v = 0; v < levels[0].n; ++v) part_out[v] = cur[v];
    if (edgecut) *edgecut = edge_cut(levels[0], cur, T);
    clk.lap("fill + cut", levels[0].n, levels[0].m());
    return PGLAMD_OK;
}

void set_weights(Graph& g, int64_t n, const int64_t* vwgt, const int64_t* vwgt2) {
    for (int c = 0; c < NC; ++c) { g.vw[c].resize(n); g.tvw[c] = 0; }
    for (int64_t i = 0; i < n; ++i) {
        g.vw[0][i] = vwgt ? std::max<int64_t>(vwgt[i], 0) : 1;
        g.vw[1][i] = vwgt2 ? std::max<int64_t>(vwgt2[i], 0) : 0;       // no second constraint: weight 0, never binding
        g.tvw[0] += g.vw[0][i]; g.tvw[1] += g.vw[1][i];
    }
}

int32_t trivial(int64_t num_nodes, int64_t nparts, int64_t* part, int64_t* edgecut, bool* done) {
    *done = true;
    if (num_nodes == 0) { if (edgecut) *edgecut = 0; return PGLAMD_OK; }
    if (nparts == 1) { std::fill(part, part + num_nodes, (int64_t)0); if (edgecut) *edgecut = 0; return PGLAMD_OK; }
    *done = false;
    return PGLAMD_OK;
}

}  // namespace

extern "C" int32_t pglamd_partition_kway2(int64_t num_nodes, const int64_t* xadj, const int64_t* adjncy, const int64_t* vwgt,
                                          const int64_t* vwgt2, const int64_t* adjwgt, int64_t nparts, double ub, double ub2,
                                          uint64_t seed, int32_t threads, int64_t* part, int64_t* edgecut) {
    if (num_nodes < 0 || nparts < 1 || (num_nodes > 0 && (!xadj || !part)))
        return pglamd::fail(PGLAMD_E_ARG, "partition_kway: bad argument");
    if (num_nodes >= INT32_MAX) return pglamd::fail(PGLAMD_E_RANGE, "partition_kway: too many nodes");
    bool done = false;
    trivial(num_nodes, nparts, part, edgecut, &done);
    if (done) return PGLAMD_OK;
    const int k = (int)std::min<int64_t>(nparts, num_nodes);
    const int64_t m = xadj[num_nodes];
    if (m > 0 && !adjncy) return pglamd::fail(PGLAMD_E_ARG, "partition_kway: adjncy NULL");
    Graph g;
    g.n = (int32_t)num_nodes;
    g.xadj.assign(xadj, xadj + num_nodes + 1);
    g.adj.resize(m); g.ew.resize(m);
    const int T = pick_threads(threads);
    run_chunks(even_bounds(m, T), [&](int, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) { g.adj[i] = (int32_t)adjncy[i]; g.ew[i] = adjwgt ? (int32_t)std::min<int64_t>(std::max<int64_t>(adjwgt[i], 1), INT32_MAX / 4) : 1; }
    });
    set_weights(g, num_nodes, vwgt, vwgt2);
    const double ubs[NC] = {ub > 1.0 ? ub : 1.03, ub2 > 1.0 ? ub2 : 1.03};
    return partition_core(g, k, ubs, seed, threads, part, edgecut);
}

extern "C" int32_t pglamd_partition_kway(int64_t num_nodes, const int64_t* xadj, const int64_t* adjncy, const int64_t* vwgt,
                                         const int64_t* adjwgt, int64_t nparts, uint64_t seed, int64_t* part, int64_t* edgecut) {
    return pglamd_partition_kway2(num_nodes, xadj, adjncy, vwgt, nullptr, adjwgt, nparts, 1.03, 1.03, seed, 0, part, edgecut);
}

// The same partitioner fed with a DIRECTED edge list: the symmetrised adjacency METIS-style partitioners expect (the reference
// warns "the input graph of metis_partition should be undirected", pgl/partition.py:61) is built here, in parallel -- every edge
// (u, v), u != v, contributes u-v and v-u with weight 1; parallel edges keep their multiplicity and are merged by the first
// contraction.  Saves the caller two 2E-element concatenations and a host index build (4.3 s of a 5 s budget at 20 M edges).
extern "C" int32_t pglamd_partition_edges(const int64_t* src, int64_t src_stride, const int64_t* dst, int64_t dst_stride,
                                          int64_t num_edges, int64_t num_nodes, const int64_t* vwgt, const int64_t* vwgt2,
                                          int64_t nparts, double ub, double ub2, uint64_t seed, int32_t threads, int64_t* part,
                                          int64_t* edgecut) {
    if (num_nodes < 0 || num_edges < 0 || nparts < 1 || (num_nodes > 0 && !part) || (num_edges > 0 && (!src || !dst)))
        return pglamd::fail(PGLAMD_E_ARG, "partition_edges: bad argument");
    if (num_nodes >= INT32_MAX || num_edges > (int64_t)1 << 40) return pglamd::fail(PGLAMD_E_RANGE, "partition_edges: graph too large");
    bool done = false;
    trivial(num_nodes, nparts, part, edgecut, &done);
    if (done) return PGLAMD_OK;
    const int k = (int)std::min<int64_t>(nparts, num_nodes);
    const int T = pick_threads(threads);
    Clock clk;
    Graph g;
    g.n = (int32_t)num_nodes;
    // degree histogram per thread block of edges, then a stable placement: deterministic whatever T is, because block t's
    // cursor of a vertex starts after the entries blocks < t contribute
    const std::vector<int64_t> eb = even_bounds(num_edges, T);
    std::vector<std::vector<int32_t>> cnt(T, std::vector<int32_t>(num_nodes, 0));
    std::atomic<int> bad{0};
    run_chunks(eb, [&](int t, int64_t b, int64_t e) {
        std::vector<int32_t>& c = cnt[t];
        for (int64_t i = b; i < e; ++i) {
            const int64_t u = src[i * src_stride], v = dst[i * dst_stride];
            if (u < 0 || v < 0 || u >= num_nodes || v >= num_nodes) { bad = 1; continue; }
            if (u == v) continue;
            ++c[u]; ++c[v];
        }
    });
    if (bad) return pglamd::fail(PGLAMD_E_RANGE, "partition_edges: node id outside [0, %lld)", (long long)num_nodes);
    g.xadj.assign(num_nodes + 1, 0);
    // cnt[t][v] becomes the cursor of block t inside v's adjacency
    run_chunks(even_bounds(num_nodes, T), [&](int, int64_t b, int64_t e) {
        for (int64_t v = b; v < e; ++v) {
            int64_t tot = 0;
            for (int t = 0; t < T; ++t) { const int32_t c = cnt[t][v]; cnt[t][v] = (int32_t)tot; tot += c; }
            g.xadj[v + 1] = tot;
        }
    });
    for (int64_t v = 0; v < num_nodes; ++v) g.xadj[v + 1] += g.xadj[v];
    const int64_t m = g.xadj[num_nodes];
    g.adj.resize(m); g.ew.assign(m, 1);
    run_chunks(eb, [&](int t, int64_t b, int64_t e) {
        std::vector<int32_t>& c = cnt[t];
        for (int64_t i = b; i < e; ++i) {
            const int64_t u = src[i * src_stride], v = dst[i * dst_stride];
            if (u == v) continue;
            g.adj[g.xadj[u] + c[u]++] = (int32_t)v;
            g.adj[g.xadj[v] + c[v]++] = (int32_t)u;
        }
    });
    cnt.clear(); cnt.shrink_to_fit();
    set_weights(g, num_nodes, vwgt, vwgt2);
    clk.lap("symmetrise", num_nodes, m);
    const double ubs[NC] = {ub > 1.0 ? ub : 1.03, ub2 > 1.0 ? ub2 : 1.03};
    return partition_core(g, k, ubs, seed, threads, part, edgecut);
}
