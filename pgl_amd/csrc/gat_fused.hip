// gat_fused.hip -- the whole attention aggregation of GATConv (pgl/nn/conv.py:331-339), forward AND
// backward, without ever materialising an [E,H] tensor:
//     alpha = send_uv(a_src, a_dst, "add") ; leaky_relu ; edge_softmax(by dst) ; dropout ;
//     out   = send_ue_recv(f, alpha, "mul", "sum")
// i.e.  out[v,h,:] = sum_{e=(u->v)} drop_e * softmax_v( leaky(a_src[u,h] + a_dst[v,h]) ) * f[u,h,:]
//
// The reference materialises four [E,H] tensors and makes ~14 passes over them.
//
// FORWARD (gat_flat_kernel<VEC, 0>): each wave walks its chunk of the dst-sorted edge stream (same
//   geometry as agg_flat_kernel: scalar index loads, lanes across the H*D columns, 8 edges in
//   flight, double buffered) carrying an ONLINE softmax state per lane (running max m, running sum
//   s, running weighted row acc -- flash-attention style rescaling).  HBM traffic = one gather of
//   f[u] (H*D*4 B) + a_src[u] (H*4 B) + 8 B of index per edge + one output row: a plain SpMM + 6 %.
//   The per-row statistics (m, s) are optionally written out ([N,H] each): they are all a backward
//   pass needs to recompute alpha_e = exp(leaky(a_src[u]+a_dst[v]) - m[v]) / s[v] on the fly.
// BACKWARD: two symmetric walks, each a plain gather pass that keeps everything it produces in registers per row
//   alpha_e   = exp(leaky(pre_e) - m[v]) / s[v],  pre_e = a_src[u] + a_dst[v]        (recomputed from the forward's statistics)
//   d pre_e   = alpha_e (drop_e <g[v], f[u]>_h - t[v,h]) * (pre_e > 0 ? 1 : slope),  t[v,h] = <g[v,h,:], out[v,h,:]>
//   MODE 2, SRC-sorted stream (row = u, gathers g[v] and the scalars of v):
//             d f[u]     = sum_{e=(u->v)} drop_e alpha_e g[v]        and        d a_src[u,h] = sum_{e=(u->v)} d pre_e
//   MODE 3, DST-sorted stream (row = v, gathers f[u] and a_src[u]):             d a_dst[v,h] = sum_{e=(u->v)} d pre_e
//   The row's own vector (f[u] resp. g[v]) rides along with every edge like the per-row scalars do (an L1 hit after the
//   row's first edge); the per-head dot product is a xor-shuffle over the lanes of a head.  No [E,H] tensor exists at
//   any point (the reference-style composition writes and re-reads four of them).
// Attention dropout is a counter-based hash of (seed, original edge id, head): forward and both
// backward kernels regenerate the same mask, nothing is stored.
// Rows longer than a chunk leave partials merged in a fixed order by gat_fixup_kernel (associative
// softmax merge, or plain sums for the backward): atomic-free, bit-reproducible.
#include "common.hpp"

#include <algorithm>

namespace pglamd {

struct alignas(16) F4 { float a, m, s, t; };       // per (node, head): a_dst, softmax max, 1 / softmax sum, <g, out>

struct GatParams {
    const float* x;                 // gathered by col: f (forward) or g = dL/dout (backward-feature)
    const float* p_col;             // per-col scalar [*,H]: a_src (forward) / a_dst (backward-feature)
    const float* p_row;             // per-row scalar [*,H]: a_dst (forward) / a_src (backward-feature)
    const float* stat_m; const float* stat_s;   // backward: softmax statistics of the dst node [N,H]
    float* out;
    float* row_max; float* row_sum;             // forward: optional statistics output [out_rows,H]
    float* out_pos; float* sum_pos;             // forward, optional (training): out restricted to the edges with pre_e > 0 [out_rows,d], their softmax mass [out_rows,H]
    const int* row; const int* col; const int* eid; const int64_t* indptr;
    float* part_head; float* part_tail;         // [n_chunks, 3, d]: acc | m | s (forward) ; [n_chunks, d] (backward)
    int* long_count; int* long_list; int* long_list2;
    int64_t out_rows, n_csr_rows;
    int E, n_chunks, chunk, n_blocks, n_grid_chunks;
    int d, H, D;
    float slope, drop_p, drop_scale;
    unsigned seed;
    // backward with attention gradients (MODE 2 / 3)
    const float* row_vec;           // the row node's own vector: f[u] (MODE 2) / g[v] (MODE 3)
    const F4* packed;               // [N,H] (a_dst, m, 1/s, t) of every node as a destination
    float* out_a;                   // [out_rows,H] d a_src (MODE 2) / d a_dst (MODE 3)
    // sddmm / additive score
    const float* f; const float* g; float* dpre;
    const float* w;                 // [H*D] score weights (additive score)
    const float* ge; float* part_w; // additive-score backward: upstream gradient [E,H]; per-chunk partials of d/dw [n_chunks, H*D]
};

template <int VEC> struct alignas(4 * VEC) FV { float v[VEC]; };

// Sum over aligned groups of `group` lanes (a power of two <= 64), result in every lane of the group.  The first four
// steps are DPP lane permutes (VALU latency, no LDS round trip): quad_perm xor-1 / xor-2, then row_half_mirror and
// row_mirror -- once every lane of a quad (resp. 8-lane half row) holds the same partial sum, ANY lane of the
// neighbouring quad (half row) supplies the missing term.  Only groups wider than 16 lanes cross DPP rows (bpermute).
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float group_sum(float v, int group) {
    float t;                                     // selects, not branches: the four DPP steps are always issued
    t = dpp_add<0xB1>(v);  v = group > 1 ? t : v;        // quad_perm [1,0,3,2]
    t = dpp_add<0x4E>(v);  v = group > 2 ? t : v;        // quad_perm [2,3,0,1]
    t = dpp_add<0x141>(v); v = group > 4 ? t : v;        // row_half_mirror
    t = dpp_add<0x140>(v); v = group > 8 ? t : v;        // row_mirror
    if (group > 16) v += __shfl_xor(v, 16);
    if (group > 32) v += __shfl_xor(v, 32);
    return v;
}

// keep-mask of attention dropout: stateless hash of (seed, original edge id, head) -> [0,1)
__device__ __forceinline__ float drop_factor(unsigned seed, int eid, int head, float p, float scale) {
    unsigned h = seed ^ ((unsigned)eid * 0x9E3779B1u) ^ ((unsigned)head * 0x85EBCA77u + 0xC2B2AE3Du);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return ((h >> 8) * (1.0f / 16777216.0f)) >= p ? scale : 0.f;
}

// MODE 0: forward (online softmax).  MODE 1: backward w.r.t. features only (additive, alpha recomputed).
// MODE 2: MODE 1 + d a_src over the src-sorted stream.  MODE 3: d a_dst over the dst-sorted stream (no feature output).
// POS (forward only): also accumulate the part of the row that comes from edges with pre_e > 0 (out_pos, and its softmax mass
// sum_pos).  With them the backward gets d a_dst[v,h] = sum_e d pre_e WITHOUT any per-edge pass:
//     sum_e alpha_e l'_e (drop_e <g,f_u> - t) = <g, out_pos> + slope <g, out - out_pos> - t (s_pos + slope (1 - s_pos)),
// l'_e = 1 where pre_e > 0 and slope elsewhere -- a per-(node, head) formula evaluated in the pack kernel.  It replaces the
// [E,H] d pre buffer of round 1 (0.64 GB written by the src-sorted walk, gathered back through a permutation by a 0.51 ms
// segment sum) by one more [N, H*D] row written per destination in the forward.
template <int VEC, int MODE, bool DROP, bool POS = false>
// (MODE 2 needs 106 VGPRs = 4 waves per SIMD against the forward's 68 = 7.  Forcing 5 / 6 waves with amdgpu_waves_per_eu makes the
//  compiler spill 11 / 32 VGPRs into the hot loop: fwd + bwd 5.09 -> 5.43 / 10.0 ms at C3, profiles/r04/gat_backward_occupancy_variants.txt.)
__global__ __launch_bounds__(kBlock) void gat_flat_kernel(GatParams p) {
#ifndef PGLAMD_GAT_ATT_U
#define PGLAMD_GAT_ATT_U 4                                  // edges per batch of the attention-gradient walks (variant builds: 3, 2)
#endif
    constexpr int U = MODE >= 2 ? PGLAMD_GAT_ATT_U : 4;
    constexpr int PW = MODE == 0 ? (POS ? 5 : 3) : MODE == 2 ? 2 : 1;   // floats per column in a partial
    constexpr bool ATT = MODE >= 2;                     // accumulates the attention-score gradient of the row node
    constexpr bool FEAT = MODE != 3;                    // accumulates / writes a feature row
    using V = FV<VEC>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    const int j0 = lane * VEC;
    const bool act = j0 < p.d;
    const int head = act ? j0 / p.D : 0;
    const int lph = p.D / VEC;                          // lanes per head (a power of two when ATT)

    if ((int)blockIdx.x >= p.n_grid_chunks) {           // zero-fill role: rows that receive no edge
        const int64_t w = ((int64_t)blockIdx.x - p.n_grid_chunks) * kWavesPerBlock + wib;
        const int64_t r0 = w * kWave;
        if (r0 >= p.out_rows) return;
        const int64_t r = r0 + lane;
        bool empty = false;
        if (r < p.out_rows) empty = (r >= p.n_csr_rows) || (p.indptr[r] == p.indptr[r + 1]);
        unsigned long long mk = __ballot(empty);
        while (mk) {
            const int l = __builtin_ctzll(mk);
            mk &= mk - 1;
            if (FEAT && act) *reinterpret_cast<V*>(p.out + (r0 + l) * p.d + j0) = V{};
            if (ATT && lane < p.H) p.out_a[(r0 + l) * p.H + lane] = 0.f;
            if (MODE == 0 && p.row_max && lane < p.H) { p.row_max[(r0 + l) * p.H + lane] = 0.f; p.row_sum[(r0 + l) * p.H + lane] = 0.f; }
            if constexpr (POS) {
                if (act) *reinterpret_cast<V*>(p.out_pos + (r0 + l) * p.d + j0) = V{};
                if (lane < p.H) p.sum_pos[(r0 + l) * p.H + lane] = 0.f;
            }
        }
        return;
    }
    const int64_t lb = xcd_swizzle(blockIdx.x, p.n_blocks);
    if (lb < 0) return;
    const int c = wave_uniform((int)lb * kWavesPerBlock + wib);
    if (c >= p.n_chunks) return;
    const cptr<int> rowp = as_const(p.row);
    const cptr<int> colp = as_const(p.col);
    const cptr<int> eidp = as_const(p.eid);
    const int e0 = chunk_cut(rowp, as_const(p.indptr), c * p.chunk, p.chunk, p.E);
    const int e1 = chunk_cut(rowp, as_const(p.indptr), c * p.chunk + p.chunk, p.chunk, p.E);
    if (e0 >= e1) return;
    const float* __restrict__ x = p.x;
    const float* __restrict__ pcol = p.p_col;
    const float* __restrict__ prow = p.p_row;
    const float* __restrict__ sm = p.stat_m;
    const float* __restrict__ ss = p.stat_s;
    const float slope = p.slope;
    constexpr bool drop = DROP;
    const bool need_eid = drop;                                   // original edge ids only feed the dropout hash

    float m = -INFINITY, s = 0.f, acc[VEC], acc_a = 0.f;
    float accp[POS ? VEC : 1], sp = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
    for (int k = 0; k < (POS ? VEC : 1); ++k) accp[k] = 0.f;
    int cur = rowp[e0];
    bool head_open = e0 > 0 && rowp[e0 - 1] == cur;

    auto store_partial = [&](bool headp) {
        if (act) {
            float* dst = (headp ? p.part_head : p.part_tail) + (int64_t)c * PW * p.d;
            V o;
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = FEAT ? acc[k] : acc_a;
            *reinterpret_cast<V*>(dst + j0) = o;
            if constexpr (MODE == 2) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) o.v[k] = acc_a;
                *reinterpret_cast<V*>(dst + p.d + j0) = o;
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) o.v[k] = m;
                *reinterpret_cast<V*>(dst + p.d + j0) = o;
#pragma unroll
                for (int k = 0; k < VEC; ++k) o.v[k] = s;
                *reinterpret_cast<V*>(dst + 2 * p.d + j0) = o;
                if constexpr (POS) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o.v[k] = accp[k];
                    *reinterpret_cast<V*>(dst + 3 * p.d + j0) = o;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o.v[k] = sp;
                    *reinterpret_cast<V*>(dst + 4 * p.d + j0) = o;
                }
            }
        }
        if (!headp && lane == 0) p.long_list[atomicAdd(p.long_count, 1)] = c;
    };
    auto store_final = [&](int r) {
        if (r >= p.out_rows || !act) return;
        if constexpr (ATT) { if ((j0 % p.D) == 0) p.out_a[(int64_t)r * p.H + head] = acc_a; }
        if constexpr (!FEAT) return;
        V o;
        const float inv = MODE == 0 ? 1.f / s : 1.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = acc[k] * inv;
        *reinterpret_cast<V*>(p.out + (int64_t)r * p.d + j0) = o;
        if (MODE == 0 && p.row_max && (j0 % p.D) == 0) { p.row_max[(int64_t)r * p.H + head] = m; p.row_sum[(int64_t)r * p.H + head] = s; }
        if constexpr (POS) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = accp[k] * inv;
            *reinterpret_cast<V*>(p.out_pos + (int64_t)r * p.d + j0) = o;
            if ((j0 % p.D) == 0) p.sum_pos[(int64_t)r * p.H + head] = sp * inv;
        }
    };
    // the row node's scalar is fetched (asynchronously, with the batch) only for edges that may OPEN a row and held
    // in a register until the next row change: no stall on the change, no per-edge reload either.
    float vr_held = act ? prow[(int64_t)cur * p.H + head] : 0.f;
    auto consume = [&](int r, int ed, float vc, float vr, float vm, float vs, float vt, const V& xv, const V& yv) {
        if (r != cur) {
            if (head_open) store_partial(true); else store_final(cur);
            head_open = false;
            cur = r;
            m = -INFINITY; s = 0.f; acc_a = 0.f; sp = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
            for (int k = 0; k < (POS ? VEC : 1); ++k) accp[k] = 0.f;
            vr_held = vr;
        }
        const float pre = vc + vr_held;
        const float l = pre > 0.f ? pre : slope * pre;
        const float df = drop ? drop_factor(p.seed, ed, head, p.drop_p, p.drop_scale) : 1.f;
        if constexpr (MODE == 0) {
            // online softmax: one of exp(m - mn), exp(l - mn) is always exp(0) = 1, so ONE exponential per edge
            const float dlt = l - m;                 // +inf on the first edge of a row (m = -inf): e^{-inf} = 0
            const float ed2 = expf(-fabsf(dlt));
            const bool up = dlt > 0.f;
            const float sc = up ? ed2 : 1.f, pe = up ? 1.f : ed2;
            s = s * sc + pe;
            const float w = pe * df;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] * sc + w * xv.v[k];
            if constexpr (POS) {
                const float pp = pre > 0.f ? 1.f : 0.f;
                sp = sp * sc + pe * pp;
                const float wp = w * pp;
#pragma unroll
                for (int k = 0; k < VEC; ++k) accp[k] = accp[k] * sc + wp * xv.v[k];
            }
            m = up ? l : m;
        } else {
            const float alpha = expf(l - vm) / vs;   // alpha_e of the destination's softmax, recomputed
            if constexpr (FEAT) {
                const float w = alpha * df;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += w * xv.v[k];
            }
        }
    };
    const float* __restrict__ rvec = p.row_vec;
    auto load_idx = [&](int e, int (&cc)[U], int (&rr)[U], int (&ee)[U]) {
#pragma unroll
        for (int i = 0; i < U; ++i) { rr[i] = rowp[e + i]; cc[i] = colp[e + i]; ee[i] = need_eid ? eidp[e + i] : 0; }
    };
    // the destination's statistics (and t) are indexed by col on the src-sorted stream, by row on the dst-sorted one
    auto load_rows = [&](const int (&cc)[U], const int (&rr)[U], V (&vx)[U], V (&vy)[U], float (&vc)[U], float (&vr)[U],
                         float (&vm)[U], float (&vs)[U], float (&vt)[U]) {
#pragma unroll
        for (int i = 0; i < U; ++i)
            if (act) {
                vx[i] = *reinterpret_cast<const V*>(x + (int64_t)cc[i] * p.d + j0);
                vc[i] = pcol[(int64_t)cc[i] * p.H + head];
                if (i == 0 || rr[i] != rr[i - 1]) vr[i] = prow[(int64_t)rr[i] * p.H + head];   // only where a row may open
                if constexpr (MODE >= 1) {
                    const int64_t si = (int64_t)(MODE == 3 ? rr[i] : cc[i]) * p.H + head;
                    vm[i] = sm[si]; vs[i] = ss[si];
                }
            }
    };

    int e = e0;
    const int n_full = (e1 - e0) / U;
    if constexpr (!ATT) {
        // three stages deep, as in agg_flat_kernel: rows of batch g consumed, rows of g+1 in flight, (scalar) indices of
        // g+2 being fetched -- the scalar-load latency is off the per-batch critical path
        int cA[U], rA[U], eA[U]; V xA[U], yA[U]; float vcA[U], vrA[U], vmA[U], vsA[U], vtA[U];
        int cB[U], rB[U], eB[U];
        if (n_full > 0) { load_idx(e, cA, rA, eA); load_rows(cA, rA, xA, yA, vcA, vrA, vmA, vsA, vtA); }
        if (n_full > 1) load_idx(e + U, cB, rB, eB);
        for (int g = 0; g < n_full; ++g) {
            int cC[U], rC[U], eC[U]; V xB[U], yB[U]; float vcB[U], vrB[U], vmB[U], vsB[U], vtB[U];
            const bool more = g + 1 < n_full, more2 = g + 2 < n_full;
            if (more) load_rows(cB, rB, xB, yB, vcB, vrB, vmB, vsB, vtB);
            if (more2) load_idx(e + 2 * U, cC, rC, eC);
    #pragma unroll
            for (int i = 0; i < U; ++i) consume(rA[i], eA[i], vcA[i], vrA[i], vmA[i], vsA[i], vtA[i], xA[i], yA[i]);
            if (more) {
    #pragma unroll
                for (int i = 0; i < U; ++i) {
                    rA[i] = rB[i]; eA[i] = eB[i]; xA[i] = xB[i]; vcA[i] = vcB[i]; vrA[i] = vrB[i];
                    if constexpr (MODE >= 1) { vmA[i] = vmB[i]; vsA[i] = vsB[i]; }
                    if constexpr (ATT) { vtA[i] = vtB[i]; yA[i] = yB[i]; }
                }
            }
            if (more2) {
    #pragma unroll
                for (int i = 0; i < U; ++i) { cB[i] = cC[i]; rB[i] = rC[i]; eB[i] = eC[i]; }
            }
            e += U;
        }
        for (; e < e1; ++e) {
            const int r = rowp[e], cc = colp[e];
            const int ed = drop ? eidp[e] : 0;
            V xv{}, yv{}; float vc = 0.f, vr = 0.f, vm = 0.f, vs = 1.f, vt = 0.f;
            if (act) {
                xv = *reinterpret_cast<const V*>(x + (int64_t)cc * p.d + j0);
                vc = pcol[(int64_t)cc * p.H + head];
                vr = prow[(int64_t)r * p.H + head];
                if constexpr (MODE >= 1) {
                    const int64_t si = (int64_t)(MODE == 3 ? r : cc) * p.H + head;
                    vm = sm[si]; vs = ss[si];
                    }
            }
            consume(r, ed, vc, vr, vm, vs, vt, xv, yv);
        }
    } else {
        // Attention-gradient walks.  The destination's four per-head scalars come PACKED (a_dst, m, s, t: one 16-byte
        // load); what belongs to the row node itself (its vector, and a_src[u] resp. the packed scalars of v) is fetched
        // only for edges that OPEN a row inside the batch and held in registers until the next row change -- on a
        // power-law graph most edges continue a row, so an edge costs two vector loads, not seven.
        const F4* __restrict__ pk = p.packed;
        float* __restrict__ dpre_out = p.dpre;
        V y_held{}; F4 p_held{}; float a_held = 0.f;
        if (act) {
            y_held = *reinterpret_cast<const V*>(rvec + (int64_t)cur * p.d + j0);
            if constexpr (MODE == 2) a_held = prow[(int64_t)cur * p.H + head];
            else p_held = pk[(int64_t)cur * p.H + head];
        }
        // Round 5: what belongs to the ROW node is no longer prefetched per batch element "in case this edge opens a row" (three to
        // seven VGPRs per element and stage: 106 VGPRs = 4 waves per SIMD) but fetched at the row change itself -- a row change
        // happens once per ~19 edges at C3, the fetch is a vector load of consecutive rows (the walk's own row order), and the 24
        // registers it frees buy two more resident waves per SIMD.
        auto open_row = [&](int r) {
            if (!act) return;
            y_held = *reinterpret_cast<const V*>(rvec + (int64_t)r * p.d + j0);
            if constexpr (MODE == 2) a_held = prow[(int64_t)r * p.H + head];
            else p_held = pk[(int64_t)r * p.H + head];
        };
        auto consume_att = [&](int r, int ed, int pos, const V& xv, const F4& pc, float ac) {
            if (r != cur) {
                if (head_open) store_partial(true); else store_final(cur);
                head_open = false;
                cur = r;
                acc_a = 0.f;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
                open_row(r);
            }
            const F4 q = MODE == 2 ? pc : p_held;                  // (a_dst, m, s, t) of the destination v
            const float pre = q.a + (MODE == 2 ? a_held : ac);     // + a_src[u]
            const float l = pre > 0.f ? pre : slope * pre;
            const float df = drop ? drop_factor(p.seed, ed, head, p.drop_p, p.drop_scale) : 1.f;
            const float alpha = __expf(l - q.m) * q.s;              // q.s = 1 / softmax sum; hardware exp2 (rel. error < 1e-6 where alpha is not negligible): gradients only
            if constexpr (FEAT) {
                const float w = alpha * df;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += w * xv.v[k];
            }
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) dot += xv.v[k] * y_held.v[k];       // <g[v], f[u]> restricted to this lane
            dot = group_sum(dot, lph);
            const float dl = alpha * (df * dot - q.t);
            const float dp = pre > 0.f ? dl : slope * dl;
            acc_a += dp;
            // [E,H] in the order of THIS walk (sequential 32-byte rows; written in original edge order it was one more random
            // line per edge, and the caller's reduction by destination gathers through a permutation either way)
            if (dpre_out && (lane & (lph - 1)) == 0 && act) dpre_out[(int64_t)pos * p.H + head] = dp;
        };
        auto load_att = [&](const int (&cc)[U], V (&vx)[U], F4 (&pc)[U], float (&ac)[U]) {
            if (!act) return;
#pragma unroll
            for (int i = 0; i < U; ++i) {
                vx[i] = *reinterpret_cast<const V*>(x + (int64_t)cc[i] * p.d + j0);
                if constexpr (MODE == 2) pc[i] = pk[(int64_t)cc[i] * p.H + head];
                else ac[i] = pcol[(int64_t)cc[i] * p.H + head];
            }
        };
        // three stages deep, as the forward walk: rows of batch g consumed, rows of g+1 in flight, (scalar) indices of g+2 being fetched
        int cA[U], rA[U], eA[U]; V xA[U]; F4 pcA[U]; float acA[U];
        int cB[U], rB[U], eB[U];
        if (n_full > 0) { load_idx(e, cA, rA, eA); load_att(cA, xA, pcA, acA); }
        if (n_full > 1) load_idx(e + U, cB, rB, eB);
        for (int g = 0; g < n_full; ++g) {
            int cC[U], rC[U], eC[U]; V xB[U]; F4 pcB[U]; float acB[U];
            const bool more = g + 1 < n_full, more2 = g + 2 < n_full;
            if (more) load_att(cB, xB, pcB, acB);
            if (more2) load_idx(e + 2 * U, cC, rC, eC);
#pragma unroll
            for (int i = 0; i < U; ++i) consume_att(rA[i], eA[i], e + i, xA[i], pcA[i], acA[i]);
            if (more) {
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    rA[i] = rB[i]; eA[i] = eB[i]; xA[i] = xB[i];
                    if constexpr (MODE == 2) pcA[i] = pcB[i]; else acA[i] = acB[i];
                }
            }
            if (more2) {
#pragma unroll
                for (int i = 0; i < U; ++i) { cB[i] = cC[i]; rB[i] = rC[i]; eB[i] = eC[i]; }
            }
            e += U;
        }
        for (; e < e1; ++e) {
            const int r = rowp[e], cc = colp[e];
            const int ed = need_eid ? eidp[e] : 0;
            V xv{}; F4 pc{}; float ac = 0.f;
            if (act) {
                xv = *reinterpret_cast<const V*>(x + (int64_t)cc * p.d + j0);
                if constexpr (MODE == 2) pc = pk[(int64_t)cc * p.H + head];
                else ac = pcol[(int64_t)cc * p.H + head];
            }
            consume_att(r, ed, e, xv, pc, ac);
        }
    }
    const bool tail_open = e1 < p.E && rowp[e1] == cur;
    if (head_open) store_partial(true);
    else if (tail_open) store_partial(false);
    else store_final(cur);
}

// merges the partials of the rows longer than a chunk (only those are split), from the work list
// the flat kernel filled.  Pass 1 (LONG = false): one wave per task, rows with <= 16 partials are
// merged right there, longer ones go to a second list.  Pass 2 (LONG = true): 1024-thread blocks, 16
// waves split one hub row's partial list, LDS combine in wave order.  Every row's own merge order is
// fixed => bit-reproducible.
constexpr int kGatFixShort = 16;
constexpr int kGatFixWaves = 16;
constexpr int kGatFixGridShort = 2048;
constexpr int kGatFixGridLong = 512;

template <int VEC, bool LONG, int MODE, bool POS = false>
__global__ __launch_bounds__(LONG ? kGatFixWaves * kWave : kBlock) void gat_fixup_kernel(GatParams p) {
    using V = FV<VEC>;
    constexpr int NW = LONG ? kGatFixWaves : 1;
    constexpr int PW = MODE == 0 ? (POS ? 5 : 3) : MODE == 2 ? 2 : 1;
    // per-wave results of the LONG pass: the vectors (acc, and accp when POS) are VEC wide per lane, the scalars (m, s, sp)
    // are one value per lane -- kept apart so that VEC = 4 with POS stays under 64 KiB of LDS
    __shared__ float red_v[LONG ? kGatFixWaves : 1][POS ? 2 : 1][LONG ? kWave * VEC : 1];
    __shared__ float red_s[LONG ? kGatFixWaves : 1][3][LONG ? kWave : 1];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    const cptr<int> rowp = as_const(p.row);
    const cptr<int64_t> ip = as_const(p.indptr);
    const int j0 = lane * VEC;
    const bool act = j0 < p.d;
    const int* list = LONG ? p.long_list2 : p.long_list;
    const int n_tasks = LONG ? p.long_count[1] : p.long_count[0];
    const int first = LONG ? (int)blockIdx.x : (int)blockIdx.x * kWavesPerBlock + wib;
    const int stride = LONG ? (int)gridDim.x : (int)gridDim.x * kWavesPerBlock;
    for (int t_id = first; t_id < n_tasks; t_id += stride) {
        const int a = wave_uniform(list[t_id]);
        const int e1 = (a + 1) * p.chunk;
        const int r = rowp[e1 - 1];
        const int64_t re = ip[r + 1];
        const int b = (int)((re - 1) / p.chunk);
        if constexpr (!LONG) {
            if (b - a > kGatFixShort) {
                if (lane == 0) p.long_list2[atomicAdd(p.long_count + 1, 1)] = a;
                continue;
            }
        }
        float m = MODE == 0 ? -INFINITY : 0.f, s = 0.f, sp = 0.f, acc[VEC], accp[POS ? VEC : 1];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
        for (int k = 0; k < (POS ? VEC : 1); ++k) accp[k] = 0.f;
        auto merge_vals = [&](const V& va, float m2, float s2, const V& vp, float sp2) {
            if constexpr (MODE == 0) {
                const float mn = fmaxf(m, m2);
                const float c1 = expf(m - mn), c2 = expf(m2 - mn);
                s = s * c1 + s2 * c2;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = acc[k] * c1 + va.v[k] * c2;
                if constexpr (POS) {
                    sp = sp * c1 + sp2 * c2;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) accp[k] = accp[k] * c1 + vp.v[k] * c2;
                }
                m = mn;
            } else {                                   // additive modes: `m` doubles as the second component (MODE 2)
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += va.v[k];
                m += m2;
                s = 1.f;
            }
        };
        auto merge = [&](const float* base) {
            if constexpr (MODE == 0) {
                V vp{}; float sp2 = 0.f;
                if constexpr (POS) { vp = *reinterpret_cast<const V*>(base + 3 * p.d + j0); sp2 = base[4 * p.d + j0]; }
                merge_vals(*reinterpret_cast<const V*>(base + j0), base[p.d + j0], base[2 * p.d + j0], vp, sp2);
            } else if constexpr (MODE == 2) merge_vals(*reinterpret_cast<const V*>(base + j0), base[p.d + j0], 1.f, V{}, 0.f);
            else merge_vals(*reinterpret_cast<const V*>(base + j0), 0.f, 1.f, V{}, 0.f);
        };
        if constexpr (!LONG) {
            if (act) {
                merge(p.part_tail + (int64_t)a * PW * p.d);
#pragma unroll 4
                for (int c = a + 1; c <= b; ++c) merge(p.part_head + (int64_t)c * PW * p.d);
            }
        } else {
            if (act)
                for (int c = a + 1 + wib; c <= b; c += NW) merge(p.part_head + (int64_t)c * PW * p.d);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                red_v[wib][0][lane * VEC + k] = acc[k];
                if constexpr (POS) red_v[wib][1][lane * VEC + k] = accp[k];
            }
            red_s[wib][0][lane] = m; red_s[wib][1][lane] = s; red_s[wib][2][lane] = sp;
            __syncthreads();
            if (wib != 0) continue;
            // wave 0: tail partial of chunk a first, then the wave results in wave order
            m = MODE == 0 ? -INFINITY : 0.f; s = 0.f; sp = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
            for (int k = 0; k < (POS ? VEC : 1); ++k) accp[k] = 0.f;
            if (act) {
                merge(p.part_tail + (int64_t)a * PW * p.d);
                for (int w = 0; w < NW; ++w) {
                    const float s2 = red_s[w][1][lane];
                    if (s2 == 0.f) continue;                     // that wave had no partial
                    V va, vp{};
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        va.v[k] = red_v[w][0][lane * VEC + k];
                        if constexpr (POS) vp.v[k] = red_v[w][1][lane * VEC + k];
                    }
                    merge_vals(va, red_s[w][0][lane], s2, vp, red_s[w][2][lane]);
                }
            }
        }
        if (!act || r >= p.out_rows) continue;
        if constexpr (MODE == 2) { if ((j0 % p.D) == 0) p.out_a[(int64_t)r * p.H + j0 / p.D] = m; }
        if constexpr (MODE == 3) { if ((j0 % p.D) == 0) p.out_a[(int64_t)r * p.H + j0 / p.D] = acc[0]; continue; }
        V o;
        const float inv = MODE == 0 ? 1.f / s : 1.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = acc[k] * inv;
        *reinterpret_cast<V*>(p.out + (int64_t)r * p.d + j0) = o;
        if (MODE == 0 && p.row_max && (j0 % p.D) == 0) { p.row_max[(int64_t)r * p.H + j0 / p.D] = m; p.row_sum[(int64_t)r * p.H + j0 / p.D] = s; }
        if constexpr (POS) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = accp[k] * inv;
            *reinterpret_cast<V*>(p.out_pos + (int64_t)r * p.d + j0) = o;
            if ((j0 % p.D) == 0) p.sum_pos[(int64_t)r * p.H + j0 / p.D] = sp * inv;
        }
    }
}

// SDDMM over a dst-sorted edge stream: out[eid[p], h] = < x[col[p], h, :], y[row[p], h, :] >.
// This is d loss / d (edge feature) of send_ue_recv(x, e, "mul", "sum") when e is [E,H,1]
// (x = node features gathered by source, y = the incoming gradient rows): the reference composes it
// from two [E,H,D] gathers; here neither is materialised.  Lanes span the H*D columns; the per-head dot
// product is a xor-shuffle reduction over the D/VEC lanes of a head (a power of two); lane 0 of each head writes.
// ADDLEAKY: out = sum_d w[h,d] * leaky(x[col] + y[row]) instead of the plain dot product (GATv2's score, pgl/nn/conv.py:421-424)
template <int VEC, bool ADDLEAKY = false>
__global__ __launch_bounds__(kBlock) void sddmm_kernel(GatParams p) {
    constexpr int U = 8;
    using V = FV<VEC>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    const int j0 = lane * VEC;
    const bool act = j0 < p.d;
    const int lph = p.D / VEC;
    const int head = act ? j0 / p.D : 0;
    const bool writer = act && (lane % lph) == 0;
    const int64_t lb = xcd_swizzle(blockIdx.x, p.n_blocks);
    if (lb < 0) return;
    const int c = wave_uniform((int)lb * kWavesPerBlock + wib);
    if (c >= p.n_chunks) return;
    const int e0 = c * p.chunk, e1 = min(e0 + p.chunk, p.E);
    const cptr<int> rowp = as_const(p.row);
    const cptr<int> colp = as_const(p.col);
    const cptr<int> eidp = as_const(p.eid);
    int cur = -1;
    V gv{}, wv{};
    if constexpr (ADDLEAKY) { if (act) wv = *reinterpret_cast<const V*>(p.w + j0); }
    auto emit = [&](int r, int ed, const V& fx) {
        if (r != cur) { cur = r; if (act) gv = *reinterpret_cast<const V*>(p.g + (int64_t)cur * p.d + j0); }
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if constexpr (ADDLEAKY) { const float z = gv.v[k] + fx.v[k]; dot += wv.v[k] * (z > 0.f ? z : p.slope * z); }
            else dot += gv.v[k] * fx.v[k];
        }
        dot = group_sum(dot, lph);
        if (writer) p.dpre[(int64_t)ed * p.H + head] = dot;
    };
    // three stages, as in agg_flat_kernel: rows of batch g consumed, rows of g+1 in flight, scalar ids of g+2 being fetched
    auto load_idx = [&](int e, int (&rr)[U], int (&cc)[U], int (&ee)[U]) {
#pragma unroll
        for (int i = 0; i < U; ++i) { rr[i] = rowp[e + i]; cc[i] = colp[e + i]; ee[i] = eidp ? eidp[e + i] : e + i; }
    };
    auto load_rows = [&](const int (&cc)[U], V (&fx)[U]) {
#pragma unroll
        for (int i = 0; i < U; ++i)
            if (act) fx[i] = *reinterpret_cast<const V*>(p.f + (int64_t)cc[i] * p.d + j0);
    };
    int e = e0;
    const int n_full = (e1 - e0) / U;
    int rA[U], cA[U], eA[U], rB[U], cB[U], eB[U];
    V fA[U];
    if (n_full > 0) { load_idx(e, rA, cA, eA); load_rows(cA, fA); }
    if (n_full > 1) load_idx(e + U, rB, cB, eB);
    for (int g = 0; g < n_full; ++g) {
        int rC[U], cC[U], eC[U];
        V fB[U];
        const bool more = g + 1 < n_full, more2 = g + 2 < n_full;
        if (more) load_rows(cB, fB);
        if (more2) load_idx(e + 2 * U, rC, cC, eC);
#pragma unroll
        for (int i = 0; i < U; ++i) emit(rA[i], eA[i], fA[i]);
        if (more) {
#pragma unroll
            for (int i = 0; i < U; ++i) { rA[i] = rB[i]; eA[i] = eB[i]; fA[i] = fB[i]; }
        }
        if (more2) {
#pragma unroll
            for (int i = 0; i < U; ++i) { rB[i] = rC[i]; cB[i] = cC[i]; eB[i] = eC[i]; }
        }
        e += U;
    }
    for (; e < e1; ++e) {
        V fx{};
        const int cc = colp[e];
        if (act) fx = *reinterpret_cast<const V*>(p.f + (int64_t)cc * p.d + j0);
        emit(rowp[e], eidp ? eidp[e] : e, fx);
    }
}

// Backward of the additive score  s[p,h] = sum_d w[h,d] * leaky(x[col_p,h,d] + y[row_p,h,d])  w.r.t. the ROW node's operand:
//     out[r,h,d]      = sum_{p in row r} ge[gi_p,h] * w[h,d] * leaky'(x[col_p,h,d] + y[r,h,d])
//     part_w[c,h*D+d] = sum_{p in chunk c} ge[gi_p,h] * leaky(x[col_p,h,d] + y[r,h,d])            (optional: d/dw partials)
// gi_p = eid[p] (or p when eid is NULL).  Same chunk walk, partials and fix-up as the additive mode of gat_flat_kernel; the
// row node's own vector is fetched only where a row may open.  Called once per orientation of the edge list.
template <int VEC>
__global__ __launch_bounds__(kBlock) void add_score_bwd_kernel(GatParams p) {
    constexpr int U = 4;
    using V = FV<VEC>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    const int j0 = lane * VEC;
    const bool act = j0 < p.d;
    const int head = act ? j0 / p.D : 0;
    if ((int)blockIdx.x >= p.n_grid_chunks) {           // zero-fill role: rows that receive no edge
        const int64_t w = ((int64_t)blockIdx.x - p.n_grid_chunks) * kWavesPerBlock + wib;
        const int64_t r0 = w * kWave;
        if (r0 >= p.out_rows) return;
        const int64_t r = r0 + lane;
        bool empty = false;
        if (r < p.out_rows) empty = (r >= p.n_csr_rows) || (p.indptr[r] == p.indptr[r + 1]);
        unsigned long long mk = __ballot(empty);
        while (mk) {
            const int l = __builtin_ctzll(mk);
            mk &= mk - 1;
            if (act) *reinterpret_cast<V*>(p.out + (r0 + l) * p.d + j0) = V{};
        }
        return;
    }
    const int64_t lb = xcd_swizzle(blockIdx.x, p.n_blocks);
    if (lb < 0) return;
    const int c = wave_uniform((int)lb * kWavesPerBlock + wib);
    if (c >= p.n_chunks) return;
    const cptr<int> rowp = as_const(p.row);
    const cptr<int> colp = as_const(p.col);
    const cptr<int> eidp = as_const(p.eid);
    const int e0 = chunk_cut(rowp, as_const(p.indptr), c * p.chunk, p.chunk, p.E);
    const int e1 = chunk_cut(rowp, as_const(p.indptr), c * p.chunk + p.chunk, p.chunk, p.E);
    V accw{};
    if (e0 < e1) {
        const float* __restrict__ x = p.x;
        const float* __restrict__ rvec = p.row_vec;
        const float* __restrict__ ge = p.ge;
        const float slope = p.slope;
        V wv{};
        if (act) wv = *reinterpret_cast<const V*>(p.w + j0);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        int cur = rowp[e0];
        bool head_open = e0 > 0 && rowp[e0 - 1] == cur;
        V y_held{};
        if (act) y_held = *reinterpret_cast<const V*>(rvec + (int64_t)cur * p.d + j0);
        auto store_row = [&](bool partial, bool headp) {
            if (!act) return;
            V o;
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = acc[k];
            if (partial) *reinterpret_cast<V*>((headp ? p.part_head : p.part_tail) + (int64_t)c * p.d + j0) = o;
            else if (cur < p.out_rows) *reinterpret_cast<V*>(p.out + (int64_t)cur * p.d + j0) = o;
        };
        auto consume = [&](int r, float gv, const V& xv, const V& yo) {
            if (r != cur) {
                store_row(head_open, true);
                head_open = false;
                cur = r;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
                y_held = yo;
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float z = xv.v[k] + y_held.v[k];
                acc[k] += gv * wv.v[k] * (z > 0.f ? 1.f : slope);
                accw.v[k] += gv * (z > 0.f ? z : slope * z);
            }
        };
        auto load_batch = [&](int e, int nb, int (&rr)[U], V (&vx)[U], float (&gg)[U], V (&yo)[U]) {
#pragma unroll
            for (int i = 0; i < U; ++i)
                if (i < nb) {
                    rr[i] = rowp[e + i];
                    const int cc = colp[e + i];
                    const int gi = eidp ? eidp[e + i] : e + i;
                    if (act) {
                        vx[i] = *reinterpret_cast<const V*>(x + (int64_t)cc * p.d + j0);
                        gg[i] = ge[(int64_t)gi * p.H + head];
                        if (i == 0 || rr[i] != rr[i - 1]) yo[i] = *reinterpret_cast<const V*>(rvec + (int64_t)rr[i] * p.d + j0);
                    }
                }
        };
        int rA[U]; V xA[U], yA[U]; float gA[U];
        int nA = min(U, e1 - e0);
        load_batch(e0, nA, rA, xA, gA, yA);
        for (int e = e0; e < e1; e += U) {
            int rB[U]; V xB[U], yB[U]; float gB[U];
            const int nB = max(0, min(U, e1 - (e + U)));
            if (nB > 0) load_batch(e + U, nB, rB, xB, gB, yB);
#pragma unroll
            for (int i = 0; i < U; ++i)
                if (i < nA) consume(rA[i], gA[i], xA[i], yA[i]);
#pragma unroll
            for (int i = 0; i < U; ++i) { rA[i] = rB[i]; xA[i] = xB[i]; yA[i] = yB[i]; gA[i] = gB[i]; }
            nA = nB;
        }
        const bool tail_open = e1 < p.E && rowp[e1] == cur;
        if (head_open) store_row(true, true);
        else if (tail_open) { store_row(true, false); if (lane == 0) p.long_list[atomicAdd(p.long_count, 1)] = c; }
        else store_row(false, false);
    }
    if (p.part_w && act) *reinterpret_cast<V*>(p.part_w + (int64_t)c * p.d + j0) = accw;     // every chunk writes (zeros if empty)
}

// t[v,h] = <g[v,h,:], out[v,h,:]> (the softmax-backward row term) and the destination-side scalars of every
// (node, head), packed so that an edge fetches them with one 16-byte load.  LANES = head_dim / 4 lanes share one
// (node, head): fully coalesced float4 reads, a DPP group sum, one 16-byte store (LANES = 0: one thread per pair).
template <int LANES>
__global__ __launch_bounds__(kBlock) void gat_pack_kernel(const float* __restrict__ a_dst, const float* __restrict__ m,
                                                         const float* __restrict__ sm, const float* __restrict__ g,
                                                         const float* __restrict__ out, int64_t n, int D,
                                                         F4* __restrict__ packed, const float* __restrict__ out_pos,
                                                         const float* __restrict__ sum_pos, float slope,
                                                         float* __restrict__ grad_a_dst) {
    // with the forward's positive-part statistics the attention-score gradient of the DESTINATION is a per-(node, head)
    // formula (see gat_flat_kernel, POS): u = <g, out_pos>, t = <g, out>  =>  d a_dst = u + slope (t - u) - t (sp + slope (1 - sp))
    auto d_adst = [&](float t, float u, float spv) { return u + slope * (t - u) - t * (spv + slope * (1.f - spv)); };
    if constexpr (LANES == 0) {
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i >= n) return;
        float t = 0.f, u = 0.f;
        for (int k = 0; k < D; ++k) { t += g[i * D + k] * out[i * D + k]; if (out_pos) u += g[i * D + k] * out_pos[i * D + k]; }
        packed[i] = F4{a_dst[i], m[i], 1.f / sm[i], t};
        if (out_pos) grad_a_dst[i] = d_adst(t, u, sum_pos[i]);
    } else {
        const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;          // float4 index into [n, D]
        const bool live = q < n * LANES;
        float t = 0.f, u = 0.f;
        if (live) {
            const float4 a = reinterpret_cast<const float4*>(g)[q], b = reinterpret_cast<const float4*>(out)[q];
            t = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            if (out_pos) {
                const float4 c = reinterpret_cast<const float4*>(out_pos)[q];
                u = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
            }
        }
        t = group_sum(t, LANES);
        if (out_pos) u = group_sum(u, LANES);
        if (live && (q % LANES) == 0) {
            const int64_t i = q / LANES;
            packed[i] = F4{a_dst[i], m[i], 1.f / sm[i], t};
            if (out_pos) grad_a_dst[i] = d_adst(t, u, sum_pos[i]);
        }
    }
}

// edges per chunk: 256 at benchmark size, shorter for small edge streams so that the launch still fills the chip (same rule and
// measurements as chunk_edges_for in aggregate.hip); PGLAMD_CHUNK pins one value (stress tests)
static int gat_chunk_edges(int64_t num_edges) {
    static const int pinned = [] {
        const char* s = getenv("PGLAMD_CHUNK");
        if (!s) return 0;
        int v = atoi(s);
        if (v < 8) v = 8;
        return v / 8 * 8;
    }();
    if (pinned) return pinned;
    return num_edges >= 12000000 ? 256 : num_edges >= 5000000 ? 128 : 64;
}

template <int VEC, int MODE, bool POS>
static int32_t launch_gat_pos(GatParams p, hipStream_t st) {
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    const int64_t zb = ceil_div(ceil_div(p.out_rows, kWave), kWavesPerBlock);
    if (p.n_chunks > 1) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
    if (p.drop_p > 0.f)
        hipLaunchKernelGGL((gat_flat_kernel<VEC, MODE, true, POS>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
    else
        hipLaunchKernelGGL((gat_flat_kernel<VEC, MODE, false, POS>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
    PGLAMD_LAUNCH_CHECK();
    if (p.n_chunks > 1) {
        hipLaunchKernelGGL((gat_fixup_kernel<VEC, false, MODE, POS>), dim3((unsigned)std::min<int64_t>(kGatFixGridShort, ceil_div(p.n_chunks, kWavesPerBlock))), dim3(kBlock), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
        hipLaunchKernelGGL((gat_fixup_kernel<VEC, true, MODE, POS>), dim3((unsigned)std::min<int64_t>(kGatFixGridLong, p.n_chunks)), dim3(kGatFixWaves * kWave), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
    }
    return PGLAMD_OK;
}

template <int VEC, int MODE>
static int32_t launch_gat(GatParams p, hipStream_t st) {
    if constexpr (MODE == 0) { if (p.out_pos) return launch_gat_pos<VEC, 0, true>(p, st); }
    return launch_gat_pos<VEC, MODE, false>(p, st);
}

// lane geometry: one 64-lane tile covers all H*D columns, VEC elements of ONE head per lane
static int gat_vec(int64_t heads, int64_t head_dim, const void* a, const void* b, const void* c, bool need_pow2) {
    const int64_t d = heads * head_dim;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
    for (int v = 1; v <= 4; v <<= 1) {
        if (d % v || head_dim % v || d / v > kWave || al % (4 * v)) continue;
        const int64_t lph = head_dim / v;
        if (need_pow2 && (lph & (lph - 1))) continue;
        return v;
    }
    return 0;
}

static void gat_setup_partials(GatParams& p, void* workspace, int pw) {
    const size_t half = align_up((size_t)p.n_chunks * pw * p.d * sizeof(float), 256);
    const size_t lst = align_up((size_t)(p.n_chunks + 64) * sizeof(int), 256);
    p.part_head = static_cast<float*>(workspace);
    p.part_tail = reinterpret_cast<float*>(static_cast<char*>(workspace) + half);
    p.long_count = reinterpret_cast<int*>(static_cast<char*>(workspace) + 2 * half);
    p.long_list = p.long_count + 64;
    p.long_list2 = reinterpret_cast<int*>(static_cast<char*>(workspace) + 2 * half + lst);
}

}  // namespace pglamd

using namespace pglamd;

extern "C" size_t pglamd_gat_aggregate_workspace_bytes(int64_t num_edges, int64_t heads, int64_t head_dim) {
    if (num_edges <= 0) return 256;
    const int64_t n_chunks = ceil_div(num_edges, gat_chunk_edges(num_edges));
    return 2 * align_up((size_t)n_chunks * 5 * heads * head_dim * sizeof(float), 256) +
           2 * align_up((size_t)(n_chunks + 64) * sizeof(int), 256) + 256;
}

extern "C" int32_t pglamd_gat_aggregate(const float* feature, const float* attn_src, const float* attn_dst, int64_t heads,
                                        int64_t head_dim, float negative_slope, float drop_p, uint32_t seed,
                                        const int32_t* row, const int32_t* col, const int32_t* eid, const int64_t* indptr,
                                        int64_t num_edges, int64_t n_csr_rows, int64_t out_rows, float* out, float* row_max,
                                        float* row_sum, float* out_pos, float* sum_pos, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    if (!out || !indptr || heads <= 0 || head_dim <= 0 || out_rows < 0 || (num_edges > 0 && (!feature || !attn_src || !attn_dst || !row || !col)))
        return fail(PGLAMD_E_ARG, "gat_aggregate: bad argument");
    if ((row_max == nullptr) != (row_sum == nullptr)) return fail(PGLAMD_E_ARG, "gat_aggregate: row_max/row_sum must both be given or both NULL");
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !eid)) return fail(PGLAMD_E_ARG, "gat_aggregate: dropout needs 0 <= p < 1 and eid");
    if (num_edges < 0 || num_edges > kMaxEdges || out_rows >= INT32_MAX) return fail(PGLAMD_E_RANGE, "gat_aggregate: sizes beyond int32 engine range");
    const int64_t d = heads * head_dim;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out_rows == 0) return PGLAMD_OK;
    if (num_edges == 0) {
        PGLAMD_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)out_rows * d * sizeof(float), st));
        if (row_max) {
            PGLAMD_HIP_CHECK(hipMemsetAsync(row_max, 0, (size_t)out_rows * heads * sizeof(float), st));
            PGLAMD_HIP_CHECK(hipMemsetAsync(row_sum, 0, (size_t)out_rows * heads * sizeof(float), st));
        }
        if (out_pos) {
            PGLAMD_HIP_CHECK(hipMemsetAsync(out_pos, 0, (size_t)out_rows * d * sizeof(float), st));
            PGLAMD_HIP_CHECK(hipMemsetAsync(sum_pos, 0, (size_t)out_rows * heads * sizeof(float), st));
        }
        return PGLAMD_OK;
    }
    if ((out_pos == nullptr) != (sum_pos == nullptr) || (out_pos && reinterpret_cast<uintptr_t>(out_pos) % 16))
        return fail(PGLAMD_E_ARG, "gat_aggregate: out_pos and sum_pos come together (out_pos 16-byte aligned)");
    const int vec = gat_vec(heads, head_dim, feature, out, workspace, false);
    if (vec == 0 || heads > kWave)
        return fail(PGLAMD_E_SHAPE, "gat_aggregate: heads*head_dim = %lld does not fit one 64-lane tile (max 256 with head_dim %% 4 == 0)", (long long)d);
    if (!workspace || workspace_bytes < pglamd_gat_aggregate_workspace_bytes(num_edges, heads, head_dim))
        return fail(PGLAMD_E_WORKSPACE, "gat_aggregate: workspace too small");
    GatParams p{};
    p.x = feature; p.p_col = attn_src; p.p_row = attn_dst; p.out = out; p.row_max = row_max; p.row_sum = row_sum;
    p.out_pos = out_pos; p.sum_pos = sum_pos;
    p.row = row; p.col = col; p.eid = eid; p.indptr = indptr;
    p.out_rows = out_rows; p.n_csr_rows = n_csr_rows; p.E = (int)num_edges;
    p.chunk = gat_chunk_edges(num_edges); p.n_chunks = (int)ceil_div(num_edges, p.chunk);
    p.d = (int)d; p.H = (int)heads; p.D = (int)head_dim; p.slope = negative_slope;
    p.drop_p = drop_p; p.drop_scale = 1.f / (1.f - drop_p); p.seed = seed;
    gat_setup_partials(p, workspace, out_pos ? 5 : 3);
    switch (vec) {
        case 1: return launch_gat<1, 0>(p, st);
        case 2: return launch_gat<2, 0>(p, st);
        default: return launch_gat<4, 0>(p, st);
    }
}

extern "C" size_t pglamd_gat_backward_workspace_bytes(int64_t num_edges, int64_t num_nodes, int64_t heads, int64_t head_dim) {
    return align_up((size_t)(num_nodes > 0 ? num_nodes : 1) * heads * sizeof(F4), 256) +
           pglamd_gat_aggregate_workspace_bytes(num_edges, heads, head_dim);
}

extern "C" int32_t pglamd_gat_backward(const float* grad_out, const float* feature, const float* attn_src,
                                       const float* attn_dst, const float* row_max, const float* row_sum, const float* out,
                                       int64_t heads, int64_t head_dim, float negative_slope, float drop_p, uint32_t seed,
                                       const int32_t* dst_row, const int32_t* dst_col, const int32_t* dst_eid,
                                       const int64_t* dst_indptr, const int32_t* src_row, const int32_t* src_col,
                                       const int32_t* src_eid, const int64_t* src_indptr, int64_t num_edges, int64_t num_nodes,
                                       float* grad_feature, float* grad_attn_src, float* grad_attn_dst, float* grad_pre,
                                       const float* out_pos, const float* sum_pos, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    if ((out_pos == nullptr) != (sum_pos == nullptr) || (out_pos && (!grad_attn_dst || reinterpret_cast<uintptr_t>(out_pos) % 16)))
        return fail(PGLAMD_E_ARG, "gat_backward: out_pos and sum_pos come together, with grad_attn_dst (out_pos 16-byte aligned)");
    if (out_pos) grad_pre = nullptr;             // d a_dst comes out of the pack kernel: neither the edge buffer nor the dst-sorted walk
    if (heads <= 0 || head_dim <= 0 || num_nodes < 0 || num_edges < 0 || !grad_feature || !grad_attn_src || (!grad_attn_dst && !grad_pre) ||
        (num_edges > 0 && (!grad_out || !feature || !attn_src || !attn_dst || !row_max || !row_sum || !out || !dst_row || !dst_col ||
                           !dst_eid || !dst_indptr || !src_row || !src_col || !src_eid || !src_indptr)))
        return fail(PGLAMD_E_ARG, "gat_backward: bad argument");
    if (drop_p < 0.f || drop_p >= 1.f) return fail(PGLAMD_E_ARG, "gat_backward: dropout needs 0 <= p < 1");
    if (num_edges > kMaxEdges || num_nodes >= INT32_MAX) return fail(PGLAMD_E_RANGE, "gat_backward: sizes beyond int32 engine range");
    const int64_t d = heads * head_dim;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (num_nodes == 0) return PGLAMD_OK;
    if (num_edges == 0) {
        PGLAMD_HIP_CHECK(hipMemsetAsync(grad_feature, 0, (size_t)num_nodes * d * sizeof(float), st));
        PGLAMD_HIP_CHECK(hipMemsetAsync(grad_attn_src, 0, (size_t)num_nodes * heads * sizeof(float), st));
        if (grad_attn_dst) PGLAMD_HIP_CHECK(hipMemsetAsync(grad_attn_dst, 0, (size_t)num_nodes * heads * sizeof(float), st));
        return PGLAMD_OK;
    }
    const int vec = gat_vec(heads, head_dim, feature, grad_out, grad_feature, true);
    if (vec == 0 || heads > kWave || reinterpret_cast<uintptr_t>(workspace) % 256 || (reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(grad_out)) % 16)
        return fail(PGLAMD_E_SHAPE, "gat_backward: heads*head_dim = %lld needs one 64-lane tile and head_dim/VEC a power of two", (long long)d);
    if (!workspace || workspace_bytes < pglamd_gat_backward_workspace_bytes(num_edges, num_nodes, heads, head_dim))
        return fail(PGLAMD_E_WORKSPACE, "gat_backward: workspace too small");
    // (0) the destination-side scalars of every (node, head), packed so that an edge fetches them with one 16-byte load
    const size_t pack_bytes = align_up((size_t)num_nodes * heads * sizeof(F4), 256);
    F4* packed = static_cast<F4*>(workspace);
    workspace = static_cast<char*>(workspace) + pack_bytes;
    {
        const int64_t pairs = num_nodes * heads;
        const int lanes = (head_dim % 4 == 0) ? (int)(head_dim / 4) : 0;
#define PGLAMD_PACK(L) hipLaunchKernelGGL(gat_pack_kernel<L>, dim3((unsigned)ceil_div(pairs * (L ? L : 1), kBlock)), dim3(kBlock), 0, st, \
                                          attn_dst, row_max, row_sum, grad_out, out, pairs, (int)head_dim, packed, out_pos, sum_pos, negative_slope, \
                                          grad_attn_dst)
        switch (lanes) {
            case 1: PGLAMD_PACK(1); break;
            case 2: PGLAMD_PACK(2); break;
            case 4: PGLAMD_PACK(4); break;
            case 8: PGLAMD_PACK(8); break;
            case 16: PGLAMD_PACK(16); break;
            default: PGLAMD_PACK(0); break;
        }
#undef PGLAMD_PACK
    }
    PGLAMD_LAUNCH_CHECK();
    GatParams p{};
    p.H = (int)heads; p.D = (int)head_dim; p.d = (int)d; p.slope = negative_slope;
    p.drop_p = drop_p; p.drop_scale = 1.f / (1.f - drop_p); p.seed = seed;
    p.E = (int)num_edges; p.chunk = gat_chunk_edges(num_edges); p.n_chunks = (int)ceil_div(num_edges, p.chunk);
    p.packed = packed;
    p.out_rows = num_nodes; p.n_csr_rows = num_nodes;
    int32_t rc;
    // (1) dst-sorted walk: row = v gathers f[u], a_src[u]; accumulates d a_dst[v].  Skipped when the caller takes
    //     d pre_e [E,H] from walk (2) instead and segment-sums it by destination itself (faster, 4*E*H bytes more memory).
    if (!grad_pre && !out_pos) {
        GatParams q = p;
        q.row = dst_row; q.col = dst_col; q.eid = dst_eid; q.indptr = dst_indptr;
        q.x = feature; q.p_col = attn_src; q.p_row = attn_dst; q.row_vec = grad_out; q.out = nullptr; q.out_a = grad_attn_dst;
        gat_setup_partials(q, workspace, 1);
        rc = vec == 1 ? launch_gat<1, 3>(q, st) : vec == 2 ? launch_gat<2, 3>(q, st) : launch_gat<4, 3>(q, st);
        if (rc != PGLAMD_OK) return rc;
    }
    // (2) src-sorted walk: row = u gathers g[v] and v's scalars; accumulates d f[u] and d a_src[u]
    p.row = src_row; p.col = src_col; p.eid = src_eid; p.indptr = src_indptr;
    p.x = grad_out; p.p_col = attn_dst; p.p_row = attn_src; p.row_vec = feature; p.out = grad_feature; p.out_a = grad_attn_src;
    p.dpre = grad_pre;
    gat_setup_partials(p, workspace, 2);
    return vec == 1 ? launch_gat<1, 2>(p, st) : vec == 2 ? launch_gat<2, 2>(p, st) : launch_gat<4, 2>(p, st);
}

extern "C" int32_t pglamd_sddmm(const float* x_by_col, const float* y_by_row, int64_t heads, int64_t head_dim,
                                const int32_t* row, const int32_t* col, const int32_t* eid, int64_t num_edges, float* out,
                                void* stream) {
    if (heads <= 0 || head_dim <= 0 || num_edges < 0 || (num_edges > 0 && (!x_by_col || !y_by_row || !row || !col || !out)))
        return fail(PGLAMD_E_ARG, "sddmm: bad argument");
    if (num_edges > kMaxEdges) return fail(PGLAMD_E_RANGE, "sddmm: sizes beyond int32 engine range");
    if (num_edges == 0) return PGLAMD_OK;
    const int vec = gat_vec(heads, head_dim, x_by_col, y_by_row, nullptr, true);
    if (vec == 0 || heads > kWave)
        return fail(PGLAMD_E_SHAPE, "sddmm: heads*head_dim = %lld needs one 64-lane tile and head_dim/VEC a power of two", (long long)(heads * head_dim));
    GatParams p{};
    p.H = (int)heads; p.D = (int)head_dim; p.d = (int)(heads * head_dim);
    p.E = (int)num_edges; p.chunk = gat_chunk_edges(num_edges); p.n_chunks = (int)ceil_div(num_edges, p.chunk);
    p.row = row; p.col = col; p.eid = eid; p.f = x_by_col; p.g = y_by_row; p.dpre = out;
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (vec) {
        case 1: hipLaunchKernelGGL(sddmm_kernel<1>, dim3((unsigned)xcd_grid(nb)), dim3(kBlock), 0, st, p); break;
        case 2: hipLaunchKernelGGL(sddmm_kernel<2>, dim3((unsigned)xcd_grid(nb)), dim3(kBlock), 0, st, p); break;
        default: hipLaunchKernelGGL(sddmm_kernel<4>, dim3((unsigned)xcd_grid(nb)), dim3(kBlock), 0, st, p); break;
    }
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

extern "C" int64_t pglamd_add_score_chunks(int64_t num_edges) { return num_edges > 0 ? ceil_div(num_edges, gat_chunk_edges(num_edges)) : 0; }

extern "C" int32_t pglamd_add_score(const float* x_by_col, const float* y_by_row, const float* w, int64_t heads, int64_t head_dim,
                                    float negative_slope, const int32_t* row, const int32_t* col, const int32_t* eid,
                                    int64_t num_edges, float* out, void* stream) {
    if (heads <= 0 || head_dim <= 0 || num_edges < 0 || (num_edges > 0 && (!x_by_col || !y_by_row || !w || !row || !col || !out)))
        return fail(PGLAMD_E_ARG, "add_score: bad argument");
    if (num_edges > kMaxEdges) return fail(PGLAMD_E_RANGE, "add_score: sizes beyond int32 engine range");
    if (num_edges == 0) return PGLAMD_OK;
    const int vec = gat_vec(heads, head_dim, x_by_col, y_by_row, w, true);
    if (vec == 0 || heads > kWave)
        return fail(PGLAMD_E_SHAPE, "add_score: heads*head_dim = %lld needs one 64-lane tile and head_dim/VEC a power of two", (long long)(heads * head_dim));
    GatParams p{};
    p.H = (int)heads; p.D = (int)head_dim; p.d = (int)(heads * head_dim); p.slope = negative_slope;
    p.E = (int)num_edges; p.chunk = gat_chunk_edges(num_edges); p.n_chunks = (int)ceil_div(num_edges, p.chunk);
    p.row = row; p.col = col; p.eid = eid; p.f = x_by_col; p.g = y_by_row; p.w = w; p.dpre = out;
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (vec) {
        case 1: hipLaunchKernelGGL((sddmm_kernel<1, true>), dim3((unsigned)xcd_grid(nb)), dim3(kBlock), 0, st, p); break;
        case 2: hipLaunchKernelGGL((sddmm_kernel<2, true>), dim3((unsigned)xcd_grid(nb)), dim3(kBlock), 0, st, p); break;
        default: hipLaunchKernelGGL((sddmm_kernel<4, true>), dim3((unsigned)xcd_grid(nb)), dim3(kBlock), 0, st, p); break;
    }
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_add_score_backward(const float* x_by_col, const float* y_by_row, const float* w, const float* grad_score,
                                             int64_t heads, int64_t head_dim, float negative_slope, const int32_t* row,
                                             const int32_t* col, const int32_t* eid, const int64_t* indptr, int64_t num_edges,
                                             int64_t num_rows, float* grad_rows, float* grad_w_partials, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    if (heads <= 0 || head_dim <= 0 || num_edges < 0 || num_rows < 0 || !grad_rows ||
        (num_edges > 0 && (!x_by_col || !y_by_row || !w || !grad_score || !row || !col || !indptr)))
        return fail(PGLAMD_E_ARG, "add_score_backward: bad argument");
    if (num_edges > kMaxEdges || num_rows >= INT32_MAX) return fail(PGLAMD_E_RANGE, "add_score_backward: sizes beyond int32 engine range");
    const int64_t d = heads * head_dim;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (num_rows == 0) return PGLAMD_OK;
    if (num_edges == 0) {
        PGLAMD_HIP_CHECK(hipMemsetAsync(grad_rows, 0, (size_t)num_rows * d * sizeof(float), st));
        return PGLAMD_OK;
    }
    const int vec = gat_vec(heads, head_dim, x_by_col, y_by_row, grad_rows, false);
    if (vec == 0 || heads > kWave || reinterpret_cast<uintptr_t>(w) % 16 || reinterpret_cast<uintptr_t>(workspace) % 16 ||
        (grad_w_partials && reinterpret_cast<uintptr_t>(grad_w_partials) % 16))
        return fail(PGLAMD_E_SHAPE, "add_score_backward: heads*head_dim = %lld does not fit one 64-lane tile", (long long)d);
    if (!workspace || workspace_bytes < pglamd_gat_aggregate_workspace_bytes(num_edges, heads, head_dim))
        return fail(PGLAMD_E_WORKSPACE, "add_score_backward: workspace too small");
    GatParams p{};
    p.H = (int)heads; p.D = (int)head_dim; p.d = (int)d; p.slope = negative_slope;
    p.E = (int)num_edges; p.chunk = gat_chunk_edges(num_edges); p.n_chunks = (int)ceil_div(num_edges, p.chunk);
    p.row = row; p.col = col; p.eid = eid; p.indptr = indptr;
    p.x = x_by_col; p.row_vec = y_by_row; p.w = w; p.ge = grad_score; p.out = grad_rows; p.part_w = grad_w_partials;
    p.out_rows = num_rows; p.n_csr_rows = num_rows;
    gat_setup_partials(p, workspace, 1);
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    const int64_t zb = ceil_div(ceil_div(p.out_rows, kWave), kWavesPerBlock);
    if (p.n_chunks > 1) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
#define PGLAMD_ASB(V)                                                                                                                   \
    do {                                                                                                                               \
        hipLaunchKernelGGL(add_score_bwd_kernel<V>, dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);                    \
        PGLAMD_LAUNCH_CHECK();                                                                                                         \
        if (p.n_chunks > 1) {                                                                                                          \
            hipLaunchKernelGGL((gat_fixup_kernel<V, false, 1>), dim3((unsigned)std::min<int64_t>(kGatFixGridShort, ceil_div(p.n_chunks, kWavesPerBlock))), dim3(kBlock), 0, st, p); \
            PGLAMD_LAUNCH_CHECK();                                                                                                     \
            hipLaunchKernelGGL((gat_fixup_kernel<V, true, 1>), dim3((unsigned)std::min<int64_t>(kGatFixGridLong, p.n_chunks)), dim3(kGatFixWaves * kWave), 0, st, p); \
            PGLAMD_LAUNCH_CHECK();                                                                                                     \
        }                                                                                                                              \
    } while (0)
    switch (vec) {
        case 1: PGLAMD_ASB(1); break;
        case 2: PGLAMD_ASB(2); break;
        default: PGLAMD_ASB(4); break;
    }
#undef PGLAMD_ASB
    return PGLAMD_OK;
}
