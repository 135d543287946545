// aggregate_flat.hpp -- the flat aggregation kernels and their typed launcher as templates, so that the (many)
// instantiations can be compiled in four translation units side by side: aggregate.hip (float + the C ABI),
// aggregate_f64.hip, aggregate_more.hip (int32, int64), aggregate_half.hip (fp16, bf16).  Design notes: see the head of aggregate.hip.
#pragma once
#include "aggregate.hpp"
#include "aggregate_group.hpp"

#include <algorithm>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace pglamd {

int narrow_max();
int32_t zero_empty_rows(const int64_t* indptr, int64_t n_csr_rows, int64_t out_rows, void* out, size_t row_bytes, hipStream_t st);

// fix-up geometry (the kernels are further down; the flat kernel classifies its tasks with kFixShort)
constexpr int kFixShort = 16;        // rows with at most this many further pieces are finished by one wave
constexpr int kFixWaves = 16;
constexpr int kFixGridShort = 2048;
constexpr int kFixGridLong = 512;
constexpr int kFixGridMergedShort = 1024;  // merged launch: blocks of kFixWaves waves, every wave of a short-role block takes its own tasks
                                           // (C2: 15 035 split rows, 211 of them hub rows -- one task per wave.  Measured: 1 024 blocks 19.6 us,
                                           //  4 096 blocks 21.1 us at C2 and 66.6 against 70.5 us at C2': more blocks than tasks cost their dispatch)


// RCLS: 0 = additive (sum / mean), 1 = min / max.   YMODE: 0 none, 1 one y per VEC group, 2 y vector,
// 3 = as 1 for NT == 1 and y rows of <= 8 elements (attention weights [E,H,1]): the 8 x ypad operand values of a batch come
//     from ONE wave-wide load (lane l: edge l / ypad, element l % ypad) and reach their lanes by ds_bpermute
// UB > 0 selects the vector-index pipeline (NT == 1, no edge operand, no per-source scale): see the main loop
// SINK = 1 (fp32 sum / mean, one tile of 64 or 128 columns, no edge operand): a finished row is NOT the result -- it is one row
//     of the left operand of a dense layer.  The wave parks finished rows in a 16-row LDS tile; a full tile is multiplied by the
//     layer's weight with v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: the reference's precision), bias and activation are
//     applied to the MFMA result and only THAT goes to memory: GCNConv's aggregate -> linear -> bias -> relu
//     (pgl/nn/conv.py:242-254) without the [N, d] intermediate's round trip through HBM, and with the matrix cores working in
//     the shadow of the row gathers (the kernel is HBM-bound; the MFMA pipe was idle).
// SS: 0 no per-source scale, 1 src_scale[col] (one random 4-byte read per edge), 2 src_scale[p] by edge POSITION (the scale of every
//     edge's source laid out along the sorted stream once per graph: 4 sequential bytes per edge)
//     A template variant, not a run-time test: the test alone cost the 256-byte-row kernel 16 SGPRs (70 -> 86: one resident workgroup
//     per CU fewer) although the code sits in the once-per-row store path.
template <typename T, int VEC, int NT, int RCLS, int YMODE, int SS = 0, bool PIPE3 = true, int UB = 0, int SINK = 0, bool TWO = false>
__global__ __launch_bounds__(kBlock) void agg_flat_kernel(AggParams p) {
    constexpr int U = 8;
    constexpr int kTileRows = 16;                      // rows per MFMA tile (v_mfma_f32_16x16x4_f32)
    constexpr int kTileStride = kWave * VEC + 4;       // floats per parked row: +4 keeps the A-operand reads bank-conflict free
    static_assert(SINK == 0 || (std::is_same_v<T, float> && NT == 1 && RCLS == 0 && YMODE == 0 && UB == 0 && (VEC == 1 || VEC == 2)),
                  "the dense sink takes fp32 rows of 64 or 128 columns");
    using V = VecT<T, VEC>;
    using A = typename AccT<T>::type;
    using VA = VecT<A, VEC>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    if ((int)blockIdx.x >= p.n_grid_chunks) {   // trailing blocks: zero-fill rows that receive no edge
        if constexpr (SINK == 1) {
            if (p.out) zero_empty_rows_role<T>(p, (int64_t)blockIdx.x - p.n_grid_chunks, lane);
            dense_empty_rows_role(p, (int64_t)blockIdx.x - p.n_grid_chunks, lane);
        } else {
            zero_empty_rows_role<T>(p, (int64_t)blockIdx.x - p.n_grid_chunks, lane);
        }
        return;
    }
    const int64_t lb = xcd_swizzle(blockIdx.x, p.n_blocks);
    if (lb < 0) return;
    const int c = wave_uniform((int)lb * kWavesPerBlock + wib);
    if (c >= p.n_chunks) return;
    const cptr<int> rowp = as_const(p.row);
    const cptr<int> colp = as_const(p.col);
    const int e0 = p.align ? chunk_cut(rowp, as_const(p.indptr), c * p.chunk, p.chunk, p.E) : c * p.chunk;
    const int e1 = p.align ? chunk_cut(rowp, as_const(p.indptr), c * p.chunk + p.chunk, p.chunk, p.E) : min(c * p.chunk + p.chunk, p.E);
    if (e0 >= e1) return;
    const cptr<int> eidp = as_const(p.eid);
    const T* __restrict__ x = static_cast<const T*>(p.x);
    const T* __restrict__ y = static_cast<const T*>(p.y);
    // TWO: second source table (rows received from peers, pgl_amd.distributed): column ids >= x_split address it.  x2 arrives
    // rebased by -x_split rows, so the address arithmetic is the same and the choice is one scalar select per edge -- compiled in
    // only where asked for: the rows of <= 320 bytes are issue-bound, and three more scalar instructions per edge cost them 25 %
    // (measured: d = 64 fp32 0.52 -> 0.69 ms with the select always on).
    const T* __restrict__ x2 = static_cast<const T*>(p.x2);
    const int xs = p.x_split;
    auto src_row = [&](int cc) -> const T* {
        if constexpr (TWO) return (cc < xs ? x : x2) + (int64_t)cc * p.ldx;
        else return x + (int64_t)cc * p.ldx;
    };
    constexpr bool has_ss = SS != 0;  // per-source scale compiled in only where asked for (keeps 16 SGPRs free otherwise)
    const bool is_max = p.is_max != 0;

    // lane -> column mapping
    int j0[NT]; bool act[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        j0[t] = (t * kWave + lane) * VEC;
        act[t] = j0[t] < p.tile_cols;
        j0[t] += p.j_base;
    }

    int yj[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) yj[t] = (YMODE == 1 || YMODE == 3) ? j0[t] / p.gy : 0;

    A acc[NT][VEC];
    auto reset = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                acc[t][k] = RCLS == 0 ? A(0) : Limits<A>::lo();
    };
    reset();
    // min / max: one v_max per element over order-reversed values for min (aggregate_group.hpp: order_flip), turned back at every store
    const bool neg = RCLS == 1 && !is_max;

    int cur = rowp[e0];
    bool head_open = e0 > 0 && rowp[e0 - 1] == cur;   // current row began in an earlier chunk

    // Everything a row store needs (output base, strides, scales, flags) is re-read from the kernarg
    // segment AT THE STORE through an opaque pointer, instead of living in ~25 SGPRs across the hot
    // loop: stores happen once per row, the freed SGPRs buy one more resident workgroup per CU.
    const cptr<AggParams> kargs = (cptr<AggParams>)__builtin_amdgcn_kernarg_segment_ptr();
    auto cold = [&]() -> cptr<AggParams> {
        cptr<AggParams> q = kargs;
        asm volatile("" : "+s"(q));      // defeats hoisting of the field loads out of the store path
        return q;
    };
    // ---- the dense sink (SINK == 1) ------------------------------------------------------------------------------------------
    float (*sink_tile)[kTileRows][kTileStride] = nullptr;
    int (*sink_rows)[kTileRows] = nullptr;
    if constexpr (SINK == 1) {                          // (declared here so that the other instantiations allocate no LDS at all)
        __shared__ float st[kWavesPerBlock][kTileRows][kTileStride];
        __shared__ int sr[kWavesPerBlock][kTileRows];
        sink_tile = st; sink_rows = sr;
    }
    int n_parked = 0;                                   // wave-uniform
    auto flush_tile = [&]() {
        if constexpr (SINK == 1) {
            const cptr<AggParams> q = cold();
            constexpr int KK = kWave * VEC / 4;         // k-steps of 4
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float a[KK];                                // A operand: lane l holds tile[l % 16][4 kk + l / 16]
            const float* trow = &sink_tile[wib][lane & 15][lane >> 4];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) a[kk] = trow[4 * kk];
            const float* __restrict__ wp = q->wp;
            const float* __restrict__ bias = q->bias;
            float* __restrict__ out2 = q->out2;
            const int dout2 = q->dout2, relu = q->act;
            const int64_t ldo2 = dout2;
            const int n_ct = dout2 >> 4;
            for (int ct = 0; ct < n_ct; ++ct) {
                float b[KK];
                const float* wt = wp + ((int64_t)ct * KK) * kWave + lane;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) b[kk] = wt[kk * kWave];
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b[kk], acc4, 0, 0, 0);
                const int colj = ct * 16 + (lane & 15);
                const float bv = bias ? bias[colj] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {           // lane l holds C[4 (l / 16) + i][l % 16]
                    const int rr = 4 * (lane >> 4) + i;
                    if (rr < n_parked) {
                        float v = acc4[i] + bv;
                        if (relu) v = v > 0.f ? v : 0.f;
                        out2[(int64_t)sink_rows[wib][rr] * ldo2 + colj] = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            n_parked = 0;
        }
    };
    auto store_partial = [&](bool head) {
        const cptr<AggParams> q = cold();
        A* dst = static_cast<A*>(head ? q->part_head : q->part_tail) + (int64_t)c * q->tile_cols;
        const int jb = q->j_base;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (act[t]) {
                VA o;
#pragma unroll
                for (int k = 0; k < VEC; ++k) o.v[k] = RCLS == 1 ? order_flip(acc[t][k], neg) : acc[t][k];
                *reinterpret_cast<VA*>(dst + (j0[t] - jb)) = o;
            }
        if (!head && lane == 0) {                           // this chunk owns the row's fix-up
            if constexpr (SINK == 0) {
                // the producer files the task under its class -- rows of <= kFixShort further pieces / hub rows -- so that both fix-up
                // roles run in ONE launch after this one (agg_fixup_merged_kernel) instead of a short pass that defers to a long pass
                const int b = (int)((as_const(q->indptr)[cur + 1] - 1) / q->chunk);
                const bool lng = b - c > kFixShort;
                (lng ? q->long_list2 : q->long_list)[atomicAdd(q->long_count + (lng ? 1 : 0), 1)] = c;
            } else {
                q->long_list[atomicAdd(q->long_count, 1)] = c;
            }
        }
    };
    // (a row stored here lies wholly inside the chunk, so the count a mean needs is the row's degree: read from indptr at
    //  the store instead of being carried -- and branched on -- at every edge: d = 64 fp32 sum 0.63 -> 0.54 ms)
    auto store_final = [&](int r) {
        const cptr<AggParams> q = cold();
        if (r >= q->out_rows) return;
        T* dst = static_cast<T*>(q->out) + (int64_t)r * q->ldo;
        const float* dsp = q->dst_scale;
        const bool is_mean = q->is_mean != 0, accumulate = q->accumulate == 1;
        float ds = 1.f;
        int64_t n = 1;
        if constexpr (RCLS == 0) {
            if (dsp) ds = as_const(dsp)[r];
            if (is_mean) { const cptr<int64_t> ipq = as_const(q->indptr); n = ipq[r + 1] - ipq[r]; }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (act[t]) {
                A ov[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    A a = RCLS == 1 ? order_flip(acc[t][k], neg) : acc[t][k];
                    if constexpr (RCLS == 0) {
                        if (is_mean) a = a / (A)n;
                        if constexpr (std::is_floating_point_v<A>) { if (dsp) a = a * (A)ds; }
                    }
                    ov[k] = a;
                }
                if (accumulate) {
                    const V old = *reinterpret_cast<const V*>(dst + j0[t]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const A ol = to_acc<T>(old.v[k]);
                        if constexpr (RCLS == 0) ov[k] = ol + ov[k];
                        else ov[k] = is_max ? (ov[k] > ol ? ov[k] : ol) : (ov[k] < ol ? ov[k] : ol);
                    }
                }
                V o;
#pragma unroll
                for (int k = 0; k < VEC; ++k) o.v[k] = from_acc<T>(ov[k]);
                if constexpr (SINK == 1) {
                    // park the (scaled) row for the MFMA tile; the aggregated row itself is stored only if the caller keeps it
                    // (training: the weight gradient needs it)
                    float* tr = &sink_tile[wib][n_parked][0];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) tr[j0[t] - q->j_base + k] = (float)ov[k];
                    if (q->out) *reinterpret_cast<V*>(dst + j0[t]) = o;
                } else {
                    *reinterpret_cast<V*>(dst + j0[t]) = o;
                }
            }
        if constexpr (SINK == 1) {
            if (lane == 0) sink_rows[wib][n_parked] = r;
            if (++n_parked == kTileRows) flush_tile();
        }
    };
    // closes row `cur` when the stream moved on to another row inside this chunk
    auto flush_mid = [&]() {
        if (head_open) store_partial(true); else store_final(cur);
        head_open = false;
    };

    // Per-source scales never touch SGPRs: lane i (i < U) of a batch fetches the column id of edge i with ONE vector
    // load in the index stage and its scale with ONE dependent vector load in the row stage; consume reads lane i back
    // (v_readlane) right where it multiplies.  What the scales do cost is memory: one random 4-byte access per edge
    // into an [N] array that the streaming feature rows keep evicting from L2 (+0.25 ms at C2 whatever the load path;
    // with the addresses pinned to one line the cost vanishes).  The host side therefore pre-multiplies narrow
    // feature rows instead (ops.aggregate) and keeps this path for wide ones.
    const float* __restrict__ sscale_v = p.src_scale;
    const int* __restrict__ col_v = p.col;
    auto scale_of = [&](int col) -> float {     // remainder loop: wave-uniform index through the vector path
        int ci = col;
        asm volatile("" : "+v"(ci));
        return sscale_v[ci];
    };
    auto load_idx = [&](int e, int (&cc)[U], int (&rr)[U], int (&yy)[U]) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            rr[i] = rowp[e + i];
            cc[i] = colp ? colp[e + i] : e + i;
            if constexpr (YMODE == 1 || YMODE == 2) yy[i] = eidp ? eidp[e + i] : e + i;
        }
    };
    // Vector loads retire in issue order (vmcnt), so the id load of batch g+2 is issued BEFORE the rows of batch g+1:
    // waiting for it one iteration later then never waits for younger row gathers.
    auto load_cl = [&](int e, int& cl) {
        if constexpr (SS == 2) cl = e + (lane & (U - 1));                 // by position: no dependence on the column ids at all
        else if constexpr (has_ss) cl = col_v ? col_v[e + (lane & (U - 1))] : e + (lane & (U - 1));
    };
    // YMODE 3: lane l carries element (l % ypad) of the operand row of edge (l / ypad) of the batch
    const int* __restrict__ eid_v = p.eid;
    const int ypad = YMODE == 3 ? p.ypad : 1;
    const int y_edge = lane / ypad, y_elem = lane % ypad;
    const bool y_lane = YMODE == 3 && y_edge < U && y_elem < (int)p.ldy;
    auto load_yl = [&](int e, int& yl) {
        if constexpr (YMODE == 3) { if (y_lane) yl = eid_v ? eid_v[e + y_edge] : e + y_edge; }
    };
    auto load_yv = [&](int yl, T& yv) {
        if constexpr (YMODE == 3) { if (y_lane) yv = y[(int64_t)yl * p.ldy + y_elem]; }
    };
    auto load_rows = [&](const int (&cc)[U], const int (&yy)[U], V (&vx)[U][NT], V (&vy)[U][NT], int cl, float& sv) {
        if constexpr (has_ss) sv = sscale_v[cl];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const T* xr = src_row(cc[i]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (act[t]) vx[i][t] = *reinterpret_cast<const V*>(xr + j0[t]);
            if constexpr (YMODE == 1) {
                const T* yr = y + (int64_t)yy[i] * p.ldy;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (act[t]) vy[i][t].v[0] = yr[yj[t]];
            } else if constexpr (YMODE == 2) {
                const T* yr = y + (int64_t)yy[i] * p.ldy;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (act[t]) vy[i][t] = *reinterpret_cast<const V*>(yr + j0[t]);
            }
        }
    };
    auto consume_one = [&](int r, float s, const V (&vx)[NT], const V (&vy)[NT], T ys = T{}) {
        if (r != cur) { flush_mid(); cur = r; reset(); }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                A m = to_acc<T>(vx[t].v[k]);
                if constexpr (std::is_floating_point_v<A>) { if (has_ss) m = m * (A)s; }
                if constexpr (YMODE == 1) m = apply_mop(m, to_acc<T>(vy[t].v[0]), p.mop);
                if constexpr (YMODE == 2) m = apply_mop(m, to_acc<T>(vy[t].v[k]), p.mop);
                if constexpr (YMODE == 3) m = apply_mop(m, to_acc<T>(ys), p.mop);
                if constexpr (RCLS == 0) acc[t][k] += m;
                else acc[t][k] = max_of(acc[t][k], order_flip(m, neg));
            }
    };
    auto lane_scale = [&](float sv, int i) -> float {
        if constexpr (has_ss) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), i));
        else return 1.f;
    };

    int e = e0;
    if constexpr (UB > 0) {
        // Vector-index pipeline (rows of <= 256 bytes, whose rate is set by rows in flight x latency, not by bytes): the row
        // and column ids of a batch of UB edges are ONE coalesced vector load each (lane l: edge l) instead of 2*UB scalar
        // loads held in SGPRs; a column id reaches the address computation through v_readlane at issue time, a row id at
        // consume time -- no index lives in an SGPR across iterations, so UB = 16 edges fit where 8 did (16..32 row gathers
        // in flight per wave) at ~50 SGPRs.  Measured at C2: d=64 fp32 0.71 -> 0.63 ms, d=32 0.74 -> 0.62, fp16 d=128 0.86 -> 0.70;
        // UB = 32 was slower again (0.72 at d=32), and 512-byte rows gain nothing (they are byte-bound), so it stops at 320 B.  Same three stages: ids of batch g+2 are issued BEFORE the rows of g+1
        // (vector loads retire in issue order, so waiting for them one iteration later never waits for younger gathers).
        static_assert(NT == 1 && YMODE == 0 && SS == 0, "vector-index pipeline: single tile, no operands");
        const int* __restrict__ row_v = p.row;
        const int* __restrict__ colv = p.col;
        const int li = lane & (UB - 1);
        auto load_iv = [&](int eb, int& rv, int& cv) { rv = row_v[eb + li]; cv = colv ? colv[eb + li] : eb + li; };
        auto load_rows_v = [&](int cv, V (&vx)[UB][1]) {
#pragma unroll
            for (int i = 0; i < UB; ++i) {
                const int cc = __builtin_amdgcn_readlane(cv, i);
                if (act[0]) vx[i][0] = *reinterpret_cast<const V*>(src_row(cc) + j0[0]);
            }
        };
        const int n_fullv = (e1 - e0) / UB;
        int rvA = 0, cvA = 0, rvB = 0, cvB = 0;
        V xvA[UB][1], wdummy[1];
        if (n_fullv > 0) load_iv(e, rvA, cvA);
        if (n_fullv > 1) load_iv(e + UB, rvB, cvB);
        if (n_fullv > 0) load_rows_v(cvA, xvA);
        for (int g = 0; g < n_fullv; ++g) {
            int rvC = 0, cvC = 0;
            V xvB[UB][1];
            const bool more = g + 1 < n_fullv, more2 = g + 2 < n_fullv;
            if (more2) load_iv(e + 2 * UB, rvC, cvC);
            if (more) load_rows_v(cvB, xvB);
#pragma unroll
            for (int i = 0; i < UB; ++i) consume_one(__builtin_amdgcn_readlane(rvA, i), 1.f, xvA[i], wdummy);
            if (more) {
                rvA = rvB; cvA = cvB;
#pragma unroll
                for (int i = 0; i < UB; ++i) xvA[i][0] = xvB[i][0];
            }
            if (more2) { rvB = rvC; cvB = cvC; }
            e += UB;
        }
    }
    // Software pipeline, three batches deep: feature rows of batch g are being consumed while the rows
    // of batch g+1 are in flight AND the (scalar) indices of batch g+2 are being fetched, so neither
    // the scalar-load latency nor the gather latency sits on the per-batch critical path.
    const int n_full = UB > 0 ? 0 : (e1 - e0) / U;
    int cA[U], rA[U], yA[U];
    int cB[U], rB[U], yB[U];
    int clA = 0, clB = 0; float svA = 1.f;
    int ylA = 0, ylB = 0; T yvA{};
    V xA[U][NT], wA[U][NT];
    if (n_full > 0) { load_cl(e, clA); load_yl(e, ylA); load_idx(e, cA, rA, yA); }
    if (n_full > 1) { load_cl(e + U, clB); load_yl(e + U, ylB); }
    if (n_full > 0) { load_yv(ylA, yvA); load_rows(cA, yA, xA, wA, clA, svA); }
    if (PIPE3 && n_full > 1) load_idx(e + U, cB, rB, yB);
    for (int g = 0; g < n_full; ++g) {
        int cC[U], rC[U], yC[U]; int clC = 0; float svB = 1.f;
        int ylC = 0; T yvB{};
        V xB[U][NT], wB[U][NT];
        const bool more = g + 1 < n_full, more2 = g + 2 < n_full;
        if (more2) { load_cl(e + 2 * U, clC); load_yl(e + 2 * U, ylC); }
        if (!PIPE3 && more) load_idx(e + U, cB, rB, yB);        // two-deep variant: fewer SGPRs, 8 workgroups per CU
        if (more) { load_yv(ylB, yvB); load_rows(cB, yB, xB, wB, clB, svB); }   // PIPE3: indices of g+1 are already in SGPRs
        if (PIPE3 && more2) load_idx(e + 2 * U, cC, rC, yC);
        if constexpr (YMODE == 3) {
            T ys[U];
#pragma unroll
            for (int i = 0; i < U; ++i) ys[i] = __shfl(yvA, i * ypad + yj[0], kWave);     // operand of edge i for this lane's column group
#pragma unroll
            for (int i = 0; i < U; ++i) consume_one(rA[i], lane_scale(svA, i), xA[i], wA[i], ys[i]);
        } else {
#pragma unroll
            for (int i = 0; i < U; ++i) consume_one(rA[i], lane_scale(svA, i), xA[i], wA[i]);
        }
        if (more) {
            svA = svB; yvA = yvB;
#pragma unroll
            for (int i = 0; i < U; ++i) {
                rA[i] = rB[i];
#pragma unroll
                for (int t = 0; t < NT; ++t) { xA[i][t] = xB[i][t]; wA[i][t] = wB[i][t]; }
            }
        }
        if (more2) {
            clB = clC; ylB = ylC;
            if constexpr (PIPE3) {
#pragma unroll
                for (int i = 0; i < U; ++i) { cB[i] = cC[i]; rB[i] = rC[i]; yB[i] = yC[i]; }
            }
        }
        e += U;
    }
    for (; e < e1; ++e) {   // remainder (< U edges): one at a time
        int r = rowp[e];
        int cc = colp ? colp[e] : e;
        float s = has_ss ? scale_of(SS == 2 ? e : cc) : 1.f;
        V vx[NT], vy[NT];
        const T* xr = src_row(cc);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (act[t]) vx[t] = *reinterpret_cast<const V*>(xr + j0[t]);
        if constexpr (YMODE != 0) {
            int yy = eidp ? eidp[e] : e;
            const T* yr = y + (int64_t)yy * p.ldy;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (act[t]) {
                    if constexpr (YMODE == 1 || YMODE == 3) vy[t].v[0] = yr[yj[t]];
                    else vy[t] = *reinterpret_cast<const V*>(yr + j0[t]);
                }
        }
        consume_one(r, s, vx, vy, (YMODE == 3 && act[0]) ? vy[0].v[0] : T{});
    }

    // the row open at the end of the chunk
    const bool tail_open = e1 < p.E && rowp[e1] == cur;
    if (head_open) store_partial(true);                 // middle or closing piece of a long row
    else if (tail_open) store_partial(false);           // first piece of a LONG row that continues
    else store_final(cur);
    if constexpr (SINK == 1) { if (n_parked > 0) flush_tile(); }
}

// ------------------------------------------------------------------------------------------------
// Fix-up: only rows LONGER than a chunk are ever split (chunk_cut), so only hub rows leave partials:
// the row's value is T[a] (+) H[a+1] (+) ... (+) H[b] with a = the chunk where it starts.  The flat
// kernel appends `a` to a work list when it writes T[a].
//   pass 1 (LONG = false): one WAVE per listed task (grid-stride over the list).  Rows with <= 16
//          partials (degree <= 17 chunks: almost all of them) are finished here, their loads issued
//          in two batches of 8; longer ones go to a second list.
//   pass 2 (LONG = true): a few 1024-thread blocks walk the second list; the 16 waves of a block split
//          one row's partial list (8 loads in flight each) and combine through LDS in wave order, so
//          a 10^5-edge hub is not one serial dependent chain.
// List order is arbitrary; every row's own combination order is fixed => bit-reproducible.
// ------------------------------------------------------------------------------------------------
// CLASSIFIED: the producer filed every task under its class already (agg_flat_kernel, SINK == 0): the short role never defers.
// `first` / `stride`: the tasks this wave (short role) or block (long role) takes.
template <typename T, int VEC, int NT, int RCLS, bool LONG, bool CLASSIFIED, typename RED>
__device__ __forceinline__ void fixup_tasks(const AggParams& p, const int first, const int stride, RED& red) {
    using A = typename AccT<T>::type;
    using V = VecT<A, VEC>;     // partials are stored in the accumulator type
    using VO = VecT<T, VEC>;
    constexpr int NW = LONG ? kFixWaves : 1;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    const cptr<int> rowp = as_const(p.row);
    const cptr<int64_t> ip = as_const(p.indptr);
    const bool is_max = p.is_max != 0;
    const A* __restrict__ ph = static_cast<const A*>(p.part_head);
    const A* __restrict__ pt = static_cast<const A*>(p.part_tail);
    auto comb = [&](A x, A y) -> A {
        if constexpr (RCLS == 0) return x + y;
        else return is_max ? (y > x ? y : x) : (y < x ? y : x);
    };
    int j0[NT]; bool act[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { j0[t] = (t * kWave + lane) * VEC; act[t] = j0[t] < p.tile_cols; }

    const int* list = LONG ? p.long_list2 : p.long_list;
    const int n_tasks = LONG ? p.long_count[1] : p.long_count[0];
    for (int t_id = first; t_id < n_tasks; t_id += stride) {
        const int a = wave_uniform(list[t_id]);
        const int e1 = (a + 1) * p.chunk;
        const int r = rowp[e1 - 1];
        const int64_t rs = ip[r], re = ip[r + 1];
        const int b = (int)((re - 1) / p.chunk);        // last chunk holding a piece of row r
        if constexpr (!LONG && !CLASSIFIED) {
            if (b - a > kFixShort) {                    // hub row: defer to the block-parallel pass
                if (lane == 0) p.long_list2[atomicAdd(p.long_count + 1, 1)] = a;
                continue;
            }
        }
        A acc[NT][VEC];
        if constexpr (!LONG) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (act[t]) {
                    const V tv = *reinterpret_cast<const V*>(pt + (int64_t)a * p.tile_cols + j0[t]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[t][k] = tv.v[k];
                }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c0 = a + 1 + half * 8;
                if (c0 > b) break;
                V v[8][NT];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c0 + u <= b) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (act[t]) v[u][t] = *reinterpret_cast<const V*>(ph + (int64_t)(c0 + u) * p.tile_cols + j0[t]);
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c0 + u <= b) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (act[t]) {
#pragma unroll
                                for (int k = 0; k < VEC; ++k) acc[t][k] = comb(acc[t][k], v[u][t].v[k]);
                            }
                    }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[t][k] = RCLS == 0 ? A(0) : (is_max ? Limits<A>::lo() : Limits<A>::hi());
            constexpr int UF = 8;
            int c = a + 1 + wib;
            for (; c + (UF - 1) * NW <= b; c += UF * NW) {
                V v[UF][NT];
#pragma unroll
                for (int u = 0; u < UF; ++u)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (act[t]) v[u][t] = *reinterpret_cast<const V*>(ph + (int64_t)(c + u * NW) * p.tile_cols + j0[t]);
#pragma unroll
                for (int u = 0; u < UF; ++u)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (act[t]) {
#pragma unroll
                            for (int k = 0; k < VEC; ++k) acc[t][k] = comb(acc[t][k], v[u][t].v[k]);
                        }
            }
            for (; c <= b; c += NW) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (act[t]) {
                        const V v = *reinterpret_cast<const V*>(ph + (int64_t)c * p.tile_cols + j0[t]);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[t][k] = comb(acc[t][k], v.v[k]);
                    }
            }
            __syncthreads();                            // previous task's readers are done with `red`
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < VEC; ++k) red[wib][(t * kWave + lane) * VEC + k] = acc[t][k];
            __syncthreads();
            if (wib != 0) continue;
            // wave 0: tail partial of chunk a first, then the wave results in wave order
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (act[t]) {
                    const V tv = *reinterpret_cast<const V*>(pt + (int64_t)a * p.tile_cols + j0[t]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        A sv = tv.v[k];
#pragma unroll
                        for (int w = 0; w < NW; ++w) sv = comb(sv, red[w][(t * kWave + lane) * VEC + k]);
                        acc[t][k] = sv;
                    }
                }
        }
        if (r >= p.out_rows) continue;
        T* dst = static_cast<T*>(p.out) + (int64_t)r * p.ldo + p.j_base;
        float ds = 1.f;
        if constexpr (RCLS == 0) { if (p.dst_scale) ds = p.dst_scale[r]; }
        if constexpr (std::is_same_v<T, float> && NT == 1 && RCLS == 0) {
            if (p.w) {
                // dense sink (aggregate -> dense layer in one launch): the finished, scaled hub row goes back into the row's own
                // tail-partial slot (nobody reads T[a] after this task); dense_hub_kernel multiplies all such rows by the layer's
                // weight 16 at a time on the matrix cores afterwards.  (As a matrix-vector product per row right here it cost
                // 0.19 ms at C2 -- one dependent L2 load per k-step -- as much as the whole matrix work of the launch.)
                float* slot = static_cast<float*>(p.part_tail) + (int64_t)a * p.tile_cols;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float v = (float)acc[0][k];
                    if (p.is_mean) v = v / (float)(re - rs);
                    if (p.dst_scale) v = v * ds;
                    if (act[0]) { slot[j0[0] + k] = v; if (p.out) dst[j0[0] + k] = v; }
                }
                __builtin_amdgcn_wave_barrier();
                continue;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (act[t]) {
                A ov[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    A v = acc[t][k];
                    if constexpr (RCLS == 0) {
                        if (p.is_mean) v = v / (A)(re - rs);
                        if constexpr (std::is_floating_point_v<A>) { if (p.dst_scale) v = v * (A)ds; }
                    }
                    ov[k] = v;
                }
                if (p.accumulate == 1) {
                    const VO old = *reinterpret_cast<const VO*>(dst + j0[t]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) ov[k] = comb(to_acc<T>(old.v[k]), ov[k]);
                }
                VO o;
#pragma unroll
                for (int k = 0; k < VEC; ++k) o.v[k] = from_acc<T>(ov[k]);
                *reinterpret_cast<VO*>(dst + j0[t]) = o;
            }
    }
}

// the two-launch form (producers that put every task on list 1: the dense sink, the grouped and the generic kernels)
template <typename T, int VEC, int NT, int RCLS, bool LONG>
__global__ __launch_bounds__(LONG ? kFixWaves * kWave : kBlock) void agg_fixup_kernel(AggParams p) {
    using A = typename AccT<T>::type;
    __shared__ A red[LONG ? kFixWaves : 1][LONG ? NT * kWave * VEC : 1];
    const int wib = wave_uniform(threadIdx.x >> 6);
    fixup_tasks<T, VEC, NT, RCLS, LONG, false>(p, LONG ? (int)blockIdx.x : (int)blockIdx.x * kWavesPerBlock + wib,
                                               LONG ? (int)gridDim.x : (int)gridDim.x * kWavesPerBlock, red);
}

// ONE launch for both classes (the flat kernel files its tasks by class): blocks [0, gl) take the hub rows (the block's 16 waves split
// one row's partial list), blocks [gl, gridDim) the short rows (every wave its own task).  The two launches cost 22 + 13 us per call at
// C2 one after the other -- the second waited for the first only because the first handed it its list.
template <typename T, int VEC, int NT, int RCLS>
__global__ __launch_bounds__(kFixWaves * kWave) void agg_fixup_merged_kernel(AggParams p, int gl) {
    using A = typename AccT<T>::type;
    __shared__ A red[kFixWaves][NT * kWave * VEC];
    const int wib = wave_uniform(threadIdx.x >> 6);
    if ((int)blockIdx.x < gl) fixup_tasks<T, VEC, NT, RCLS, true, true>(p, (int)blockIdx.x, gl, red);
    else fixup_tasks<T, VEC, NT, RCLS, false, true>(p, ((int)blockIdx.x - gl) * kFixWaves + wib, ((int)gridDim.x - gl) * kFixWaves, red);
}

// Zero-fills output rows that receive no edge: rows r < n_csr_rows with indptr[r]==indptr[r+1],
// and rows in [n_csr_rows, out_rows).  One wave inspects 64 rows (coalesced indptr read).
template <typename W>
__global__ __launch_bounds__(kBlock) void zero_empty_rows_kernel(const int64_t* __restrict__ indptr,
                                                                int64_t n_csr_rows, int64_t out_rows,
                                                                W* __restrict__ out, int64_t row_words) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t r0 = w * kWave;
    if (r0 >= out_rows) return;
    const int64_t r = r0 + lane;
    bool empty = false;
    if (r < out_rows) empty = (r >= n_csr_rows) || (indptr[r] == indptr[r + 1]);
    unsigned long long m = __ballot(empty);
    W z{};
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        W* dst = out + (r0 + l) * row_words;
        for (int64_t j = lane; j < row_words; j += kWave) dst[j] = z;
    }
}

// Catch-all: any dtype handled as T, any trailing-dim broadcast (gx, gy), any width.
// One wave per destination row, lanes stride over output columns, edges serial.
template <typename T>
__global__ __launch_bounds__(kBlock) void agg_generic_kernel(AggParams p, int gx) {
    using A = typename AccT<T>::type;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= p.out_rows) return;
    const T* __restrict__ x = static_cast<const T*>(p.x);
    const T* __restrict__ y = static_cast<const T*>(p.y);
    T* out = static_cast<T*>(p.out) + r * p.ldo;
    const int64_t n_csr = p.n_csr_rows;
    int64_t s = 0, t = 0;
    if (r < n_csr) { s = p.indptr[r]; t = p.indptr[r + 1]; }
    const bool additive = !(p.is_max == 1 || p.is_max == 2);
    for (int j = lane; j < p.tile_cols; j += kWave) {
        A acc = A(0);
        for (int64_t q = s; q < t; ++q) {
            const int cc = p.col ? p.col[q] : (int)q;
            A m = to_acc<T>((cc < p.x_split ? x : static_cast<const T*>(p.x2))[(int64_t)cc * p.ldx + j / gx]);
            if (y) {
                const int64_t yy = p.eid ? p.eid[q] : q;
                m = apply_mop(m, to_acc<T>(y[yy * p.ldy + j / p.gy]), p.mop);
            }
            if (additive) acc += m;
            else if (q == s) acc = m;
            else if (p.is_max == 1) acc = m > acc ? m : acc;
            else acc = m < acc ? m : acc;
        }
        if (additive && p.is_mean && t > s) acc = acc / (A)(t - s);
        if (p.accumulate == 1) {
            const A ol = to_acc<T>(out[j]);
            if (t > s) out[j] = from_acc<T>(additive ? ol + acc : (p.is_max == 1 ? (acc > ol ? acc : ol) : (acc < ol ? acc : ol)));
        } else if (p.accumulate == 2) {
            if (t > s) out[j] = from_acc<T>(acc);
        } else {
            out[j] = from_acc<T>(acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
template <typename T> const char* type_name() {
    return std::is_same_v<T, float> ? "float" : std::is_same_v<T, double> ? "double" : std::is_same_v<T, int32_t> ? "int" :
           std::is_same_v<T, int64_t> ? "long" : std::is_same_v<T, __half> ? "__half" : "__hip_bfloat16";
}
template <typename T> std::string kernel_name(int vec, int nt, int rcls, int ymode) {
    char b[96];
    snprintf(b, sizeof(b), "agg_flat_kernel<%s, %d, %d, %d, %d>", type_name<T>(), vec, nt, rcls, ymode);
    return b;
}

template <typename T, int VEC, int NT, int RCLS, int YMODE>
int32_t launch_flat(AggParams p, hipStream_t st) {
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    // consecutive chunks on ONE XCD (shared L2 for partition-ordered graphs) -- unless the caller's rows are ordered by something that
    // correlates with their LENGTH (HaloPlan(row_order="peers"): rows grouped by the set of peers that read them, i.e. by degree
    // class): a blocked mapping then gives one XCD all the store-heavy short-row chunks (measured: 1.11 vs 1.00 ms per rank at
    // C2' / P = 8).  The caller says so PER CALL (pglamd_aggregate_ext flags & PGLAMD_AGG_DEAL_CHUNKS; PGLAMD_XCD_SWIZZLE=0 forces it
    // for experiments): the chunks are then dealt round the XCDs.  No process-wide state is involved.
    static const bool env_off = [] { const char* sw = getenv("PGLAMD_XCD_SWIZZLE"); return sw && sw[0] == '0'; }();
    p.n_blocks = (env_off || p.deal_chunks) ? -(int)nb : (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    const int64_t zb = p.accumulate ? 0 : ceil_div(ceil_div(p.out_rows, kWave), kWavesPerBlock);
    const bool fixups = needs_fixups(p);
    if (fixups) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool profiling = prof().on.load(std::memory_order_relaxed);
    if (profiling) {
        { std::lock_guard<std::mutex> lk(prof().mu); prof().last_kernel = kernel_name<T>(VEC, NT, RCLS, YMODE); }
        PGLAMD_HIP_CHECK(hipEventCreate(&e0));
        PGLAMD_HIP_CHECK(hipEventCreate(&e1));
        PGLAMD_HIP_CHECK(hipEventRecord(e0, st));
    }
    constexpr bool can_scale = RCLS == 0 && std::is_floating_point_v<typename AccT<T>::type>;
    // 256-byte rows (d=64 fp32, d=128 fp16) are occupancy-bound: the two-deep pipeline (69 SGPRs, 8
    // workgroups/CU) measured 4 % faster there; everywhere else the three-deep one wins (up to 20 % on [E,8]).
    const size_t row_bytes = (size_t)p.tile_cols * sizeof(T);
    const bool two = p.x_split != INT32_MAX;
#define PGLAMD_LAUNCH_FLAT(...)                                                                                                          \
    do {                                                                                                                                 \
        const dim3 grid_((unsigned)(p.n_grid_chunks + zb));                                                                              \
        if (two) hipLaunchKernelGGL((agg_flat_kernel<T, VEC, NT, RCLS, YMODE, __VA_ARGS__, 0, true>), grid_, dim3(kBlock), 0, st, p);      \
        else hipLaunchKernelGGL((agg_flat_kernel<T, VEC, NT, RCLS, YMODE, __VA_ARGS__, 0, false>), grid_, dim3(kBlock), 0, st, p);         \
        PGLAMD_LAUNCH_CHECK();                                                                                                           \
    } while (0)
    if constexpr (NT == 1 && YMODE == 0) {
        static const int vidx_max = [] { const char* e = getenv("PGLAMD_VIDX_BYTES"); return e ? atoi(e) : 320; }();
        if ((int)row_bytes <= vidx_max && !p.src_scale) {
            PGLAMD_LAUNCH_FLAT(false, true, 16);
            goto launched;
        }
        if (row_bytes >= 192 && row_bytes <= 320 && !p.src_scale) {
            PGLAMD_LAUNCH_FLAT(false, false, 0);
            goto launched;
        }
    }
    if constexpr (can_scale) {
        if (p.src_scale) {           // (a second table excludes src_scale: aggregate_typed refuses the combination)
            bool by_pos = false;
            if constexpr (std::is_same_v<T, float> && YMODE == 0) {
                if (p.ss_by_pos) {
                    by_pos = true;
                    if (two) hipLaunchKernelGGL((agg_flat_kernel<T, VEC, NT, RCLS, YMODE, 2, true, 0, 0, true>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
                    else hipLaunchKernelGGL((agg_flat_kernel<T, VEC, NT, RCLS, YMODE, 2>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
                }
            }
            if (!by_pos) hipLaunchKernelGGL((agg_flat_kernel<T, VEC, NT, RCLS, YMODE, 1>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
            PGLAMD_LAUNCH_CHECK();
        } else {
            PGLAMD_LAUNCH_FLAT(false, true, 0);
        }
    } else {
        PGLAMD_LAUNCH_FLAT(false, true, 0);
    }
#undef PGLAMD_LAUNCH_FLAT
launched:
    if (profiling) {
        PGLAMD_HIP_CHECK(hipEventRecord(e1, st));
        std::lock_guard<std::mutex> lk(prof().mu);
        prof().ev.emplace_back(e0, e1);
    }
    if (fixups) {
        const int gl = (int)std::min<int64_t>(kFixGridLong, p.n_chunks), gs = (int)std::min<int64_t>(kFixGridMergedShort, ceil_div(p.n_chunks, kFixWaves));
        hipLaunchKernelGGL((agg_fixup_merged_kernel<T, VEC, NT, RCLS>), dim3((unsigned)(gl + gs)), dim3(kFixWaves * kWave), 0, st, p, gl);
    }
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

template <typename T> constexpr int32_t dtype_code() {
    return std::is_same_v<T, float> ? PGLAMD_F32 : std::is_same_v<T, double> ? PGLAMD_F64 : std::is_same_v<T, int32_t> ? PGLAMD_I32 :
           std::is_same_v<T, int64_t> ? PGLAMD_I64 : std::is_same_v<T, __half> ? PGLAMD_F16 : PGLAMD_BF16;
}

template <typename T, int VEC, int NT>
int32_t dispatch_mode(const AggParams& p, int rcls, int ymode, hipStream_t st, bool* handled) {
    *handled = true;
    if (rcls == 0) {
        if (ymode == 0) return launch_flat<T, VEC, NT, 0, 0>(p, st);
        if constexpr (std::is_floating_point_v<T>) {      // fp32 / fp64 only: 16-bit and integer operands take the generic path
            if constexpr (NT == 1) { if (ymode == 1 && p.ypad > 0) return launch_flat<T, VEC, NT, 0, 3>(p, st); }
            if (ymode == 1) return launch_flat<T, VEC, NT, 0, 1>(p, st);
            if (ymode == 2) return launch_flat<T, VEC, NT, 0, 2>(p, st);
        }
    } else if (ymode == 0) {
        return launch_flat<T, VEC, NT, 1, 0>(p, st);
    }
    *handled = false;
    return PGLAMD_OK;
}

// picks (VEC, NT) for one column tile of width w; widths are capped by max_tile_cols<T>()
template <typename T>
int32_t dispatch_shape(const AggParams& p, int vec, int rcls, int ymode, hipStream_t st, bool* handled) {
    const int w = p.tile_cols;
    if constexpr (sizeof(T) == 2) {
        if (vec >= 8) {
            if (w <= 512) return dispatch_mode<T, 8, 1>(p, rcls, ymode, st, handled);
            return dispatch_mode<T, 8, 2>(p, rcls, ymode, st, handled);
        }
        if (vec == 4) {
            if (w <= 256) return dispatch_mode<T, 4, 1>(p, rcls, ymode, st, handled);
            if (w <= 512) return dispatch_mode<T, 4, 2>(p, rcls, ymode, st, handled);
            return dispatch_mode<T, 4, 4>(p, rcls, ymode, st, handled);
        }
        if (vec == 2) {
            if (w <= 128) return dispatch_mode<T, 2, 1>(p, rcls, ymode, st, handled);
            if (w <= 256) return dispatch_mode<T, 2, 2>(p, rcls, ymode, st, handled);
            return dispatch_mode<T, 2, 4>(p, rcls, ymode, st, handled);
        }
        if (w <= 64) return dispatch_mode<T, 1, 1>(p, rcls, ymode, st, handled);
        if (w <= 128) return dispatch_mode<T, 1, 2>(p, rcls, ymode, st, handled);
        return dispatch_mode<T, 1, 4>(p, rcls, ymode, st, handled);
    } else if constexpr (sizeof(T) == 4) {
        if (vec >= 4) {
            if (w <= 256) return dispatch_mode<T, 4, 1>(p, rcls, ymode, st, handled);
            if (w <= 512) return dispatch_mode<T, 4, 2>(p, rcls, ymode, st, handled);
            return dispatch_mode<T, 4, 4>(p, rcls, ymode, st, handled);
        }
        if (vec == 2) {
            if (w <= 128) return dispatch_mode<T, 2, 1>(p, rcls, ymode, st, handled);
            if (w <= 256) return dispatch_mode<T, 2, 2>(p, rcls, ymode, st, handled);
            return dispatch_mode<T, 2, 4>(p, rcls, ymode, st, handled);
        }
        if (w <= 64) return dispatch_mode<T, 1, 1>(p, rcls, ymode, st, handled);
        if (w <= 128) return dispatch_mode<T, 1, 2>(p, rcls, ymode, st, handled);
        return dispatch_mode<T, 1, 4>(p, rcls, ymode, st, handled);
    } else {
        if (vec >= 2) {
            if (w <= 128) return dispatch_mode<T, 2, 1>(p, rcls, ymode, st, handled);
            if (w <= 256) return dispatch_mode<T, 2, 2>(p, rcls, ymode, st, handled);
            return dispatch_mode<T, 2, 4>(p, rcls, ymode, st, handled);
        }
        if (w <= 64) return dispatch_mode<T, 1, 1>(p, rcls, ymode, st, handled);
        if (w <= 128) return dispatch_mode<T, 1, 2>(p, rcls, ymode, st, handled);
        return dispatch_mode<T, 1, 4>(p, rcls, ymode, st, handled);
    }
}

template <typename T> int max_vec() { return sizeof(T) == 2 ? 8 : sizeof(T) == 4 ? 4 : 2; }
// columns one launch covers: 64 lanes x VEC x NT(max)
template <typename T> int max_tiles(int vec) { return (sizeof(T) == 2 && vec == 8) ? 2 : 4; }

// What pglamd_aggregate_ext adds to pglamd_aggregate (include/pgl_amd.h): a second source table for column ids >= x_split
// (x2 = NULL: none), the indptr that decides which rows the zero-fill clears (NULL: the launch's own), and the longest row of
// the index as a hint (0 = unknown) that lets the launcher skip the split-row fix-up when no row can be split, and row strides
// (elements; 0 = dense) of x / x2 and out: a launch may read and write a COLUMN BLOCK of wider matrices (the column-pipelined
// halo exchange aggregates columns [0, d/2) while columns [d/2, d) are still on the wire).
struct AggExtra {
    const void* x2 = nullptr; int64_t x_split = 0; const int64_t* zero_indptr = nullptr; int64_t max_row_edges = 0;
    int64_t ldx = 0, ldo = 0;
    int32_t flags = 0;                           // PGLAMD_AGG_DEAL_CHUNKS: chunks dealt round the XCDs instead of in blocks (per call)
};

// argument list of aggregate_typed<T> for the explicit instantiations (aggregate*.hip)
#define PGLAMD_AGG_ARGS const void*, int64_t, const void*, int64_t, const int32_t*, const int32_t*, const int32_t*, const int64_t*, \
                        int64_t, int64_t, int64_t, int64_t, int32_t, int32_t, const float*, const float*, int, void*, void*, size_t, \
                        const AggExtra&, hipStream_t

template <typename T>
int32_t aggregate_typed(const void* x, int64_t dx, const void* y, int64_t dy, const int32_t* eid,
                               const int32_t* row, const int32_t* col, const int64_t* indptr, int64_t E,
                               int64_t n_csr_rows, int64_t out_rows, int64_t dout, int32_t mop, int32_t rop,
                               const float* src_scale, const float* dst_scale, int accumulate, void* out, void* ws,
                               size_t ws_bytes, const AggExtra& ex, hipStream_t st) {
    int32_t rc = PGLAMD_OK;
    if (accumulate && rop == PGLAMD_MEAN)
        return fail(PGLAMD_E_ARG, "aggregate: accumulate with MEAN is undefined (use SUM with dst_scale = 1/degree)");
    const int64_t* zip = ex.zero_indptr ? ex.zero_indptr : indptr;
    const int64_t ldx = ex.ldx ? ex.ldx : dx, ldo = ex.ldo ? ex.ldo : dout;
    if (ldx < dx || ldo < dout) return fail(PGLAMD_E_SHAPE, "aggregate_ext: row stride shorter than the row (ldx %lld < %lld or ldout %lld < %lld)",
                                            (long long)ldx, (long long)dx, (long long)ldo, (long long)dout);
    if (E == 0) {
        if (accumulate) return PGLAMD_OK;
        if (ldo != dout) return fail(PGLAMD_E_ARG, "aggregate_ext: an index without edges cannot zero-fill a strided output");
        return zero_empty_rows(zip, n_csr_rows, out_rows, out, (size_t)dout * sizeof(T), st);
    }

    AggParams p{};
    p.deal_chunks = (ex.flags & PGLAMD_AGG_DEAL_CHUNKS) ? 1 : 0;
    // A one-value edge operand given IN THE ORDER OF THE SORTED STREAM (eid NULL) and multiplied into fp32 rows that are summed
    // is a per-edge scale read sequentially: it rides in the flat kernel's scale slot (SS = 2: one coalesced 4-byte load per
    // edge) instead of the general edge-operand path.  This is how GCN's source-side degree norm is applied (pgl/nn/conv.py:242):
    // norm[col[p]] laid out once per graph, rather than a pass over [N, d] per layer or a random 4-byte read per edge.
    if constexpr (std::is_same_v<T, float>) {
        if (y && dy == 1 && !eid && mop == PGLAMD_MUL && (rop == PGLAMD_SUM || rop == PGLAMD_MEAN) && !src_scale && dout == dx &&
            (size_t)dout * sizeof(T) > 128) {        // (with or without a second source table: the scale rides by edge POSITION, not by column id)
            src_scale = static_cast<const float*>(y);
            p.ss_by_pos = 1;
            y = nullptr; dy = 0;
        }
    }
    p.x = x; p.y = y; p.out = out; p.row = row; p.col = col; p.eid = eid; p.indptr = indptr;
    p.zero_indptr = zip;
    // the second table is rebased by -x_split rows here, so that the kernels address both tables with the same column id
    p.x2 = ex.x2 ? static_cast<const char*>(ex.x2) - ex.x_split * ldx * (int64_t)sizeof(T) : x;
    p.x_split = ex.x2 ? (int)ex.x_split : INT32_MAX;
    p.max_row_edges = (int)std::min<int64_t>(ex.max_row_edges, INT32_MAX);
    if (ex.x2 && ((src_scale && !p.ss_by_pos) || dout != dx))
        return fail(PGLAMD_E_ARG, "aggregate_ext: a second source table excludes a per-NODE src_scale and source-side broadcasting");
    p.src_scale = src_scale; p.dst_scale = dst_scale;
    p.ldx = ldx; p.ldy = dy; p.ldo = ldo; p.out_rows = out_rows; p.n_csr_rows = n_csr_rows; p.E = (int)E;
    p.mop = mop; p.is_mean = rop == PGLAMD_MEAN; p.is_max = rop == PGLAMD_MAX; p.accumulate = accumulate;
    const int rcls = (rop == PGLAMD_SUM || rop == PGLAMD_MEAN) ? 0 : 1;
    const int gx = (int)(dout / dx);
    const int gy = y ? (int)(dout / dy) : 1;
    p.gy = gy;

    // fast path eligibility + lane geometry: VEC elements per lane.  Candidates are limited by
    // divisibility / pointer alignment; among them take the one that needs the fewest 64-lane
    // tiles and, for equal tiles, keeps the most lanes busy (d=128 fp32 -> VEC 2: 64 x 8 B).
    bool fast = gx == 1;
    int vmax = max_vec<T>();
    const uintptr_t align_bits = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                                 (y && gy == 1 ? reinterpret_cast<uintptr_t>(y) : 0) |
                                 reinterpret_cast<uintptr_t>(ws);
    while (vmax > 1 && (dout % vmax != 0 || ldx % vmax != 0 || ldo % vmax != 0 || align_bits % (vmax * sizeof(T)) != 0)) vmax >>= 1;
    int ymode = 0;
    if (y) {
        if (gy == 1) ymode = 2;
        else { ymode = 1; if (dy <= 8) { int pd = 1; while (pd < dy) pd <<= 1; p.ypad = pd; } while (vmax > 1 && gy % vmax != 0) vmax >>= 1; }
    }
    int vec = vmax;
    {
        int64_t best_tiles = -1, best_lanes = -1;
        for (int v = vmax; v >= 1; v >>= 1) {
            const int64_t lanes = dout / v, tiles = ceil_div(lanes, kWave);
            const int64_t busy = lanes < kWave ? lanes : kWave;
            if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && busy > best_lanes)) {
                best_tiles = tiles; best_lanes = busy; vec = v;
            }
        }
        static const int forced = [] { const char* e = getenv("PGLAMD_VEC"); return e ? atoi(e) : 0; }();
        if (forced >= 1 && forced <= vmax && (forced & (forced - 1)) == 0) vec = forced;
    }
    p.zvec = vec > 4 ? 4 : vec;
    { static const int al = [] { const char* e = getenv("PGLAMD_ALIGN"); return e ? atoi(e) : 1; }(); p.align = al; }
    if ((src_scale || dst_scale) && (rcls != 0 || !std::is_floating_point_v<typename AccT<T>::type>))
        return fail(PGLAMD_E_ARG, "src_scale/dst_scale need a floating dtype and sum/mean");

    if (fast) {
        const int K = chunk_edges_for(E);
        p.chunk = K;
        p.n_chunks = (int)ceil_div(E, K);
        const int max_cols = kWave * vec * max_tiles<T>(vec);
        const int64_t tile_full = dout < max_cols ? dout : max_cols;
        const size_t half = align_up((size_t)p.n_chunks * tile_full * sizeof(typename AccT<T>::type), 256);
        const size_t lst = align_up((size_t)(p.n_chunks + 64) * sizeof(int), 256);
        const size_t need = 2 * half + 2 * lst;
        if (!ws || ws_bytes < need) return fail(PGLAMD_E_WORKSPACE, "aggregate: workspace %zu < %zu", ws_bytes, need);
        p.part_head = ws;
        p.part_tail = static_cast<char*>(ws) + half;
        p.long_count = reinterpret_cast<int*>(static_cast<char*>(ws) + 2 * half);
        p.long_list = p.long_count + 64;
        p.long_list2 = reinterpret_cast<int*>(static_cast<char*>(ws) + 2 * half + lst);
        // rows of 64..128 bytes without an edge operand: several edges per wave instruction (aggregate_group.hpp).  Below
        // 64 bytes the lane-per-edge kernel keeps sum / mean (equal at d = 16 fp32, better below); min / max (0.45 -> 0.38 ms
        // at d = 16) and the shapes it does not cover (fp16 d = 17..32: 0.63 -> 0.38) come here from 32 bytes up.
        {
            const int64_t rb = (int64_t)((size_t)dout * sizeof(T));
            const bool narrow_ok = dout <= narrow_max() && (size_t)dout * sizeof(typename AccT<T>::type) <= 64u;
            const int64_t gmin = (rcls == 1 || !narrow_ok) ? std::min<int64_t>(32, group_min_bytes()) : group_min_bytes();
            if (ymode == 0 && !src_scale && rb > gmin && rb <= group_row_bytes()) {
                AggParams q = p;
                q.j_base = 0; q.tile_cols = (int)dout;
                bool handled = false;
                rc = launch_group<T>(q, vmax, rcls, dtype_code<T>(), ws, ws_bytes, st, &handled);
                if (rc != PGLAMD_OK || handled) return rc;
            }
        }
        // measured at C2 sizes: the lane-per-edge kernel wins up to 32 B of accumulator per row for every reduce op
        // (2.6-3.4x at d <= 8 fp32) and up to 64 B for sum / mean (1.3x at d = 16 fp32)
        if (dout <= narrow_max() && (size_t)dout * sizeof(typename AccT<T>::type) <= 64u) {
            AggParams q = p;
            const int nk = std::max(K, narrow_chunk_edges());                   // fewer, longer chunks: the carved arrays still fit
            q.chunk = nk; q.n_chunks = (int)ceil_div(E, nk);
            q.j_base = 0; q.tile_cols = (int)dout;
            const size_t lv = std::min<size_t>(16, (size_t)dout * sizeof(T));
            q.narrow_vec = (lv & (lv - 1)) == 0 && ((size_t)ldx * sizeof(T)) % lv == 0 && ((size_t)ldo * sizeof(T)) % lv == 0 &&
                           (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                                                    (y && dy == dout ? reinterpret_cast<uintptr_t>(y) : 0)) % lv == 0;
            bool handled = false;
            rc = launch_narrow(q, dtype_code<T>(), rcls, y ? dy : 0, st, &handled);
            if (rc != PGLAMD_OK || handled) return rc;
        }
        for (int64_t jb = 0; jb < dout; jb += max_cols) {
            p.j_base = (int)jb;
            p.tile_cols = (int)((dout - jb) < max_cols ? (dout - jb) : max_cols);
            bool handled = false;
            rc = dispatch_shape<T>(p, vec, rcls, ymode, st, &handled);
            if (rc != PGLAMD_OK) return rc;
            if (!handled) { fast = false; break; }
        }
        if (fast) return PGLAMD_OK;
    }
    // generic fallback: rewrites every row < out_rows (rows without edges get 0)
    p.tile_cols = (int)dout; p.j_base = 0;
    p.is_max = rop == PGLAMD_MAX ? 1 : rop == PGLAMD_MIN ? 2 : 0;
    if (src_scale || dst_scale) return fail(PGLAMD_E_SHAPE, "scales unsupported with this broadcast pattern");
    hipLaunchKernelGGL(agg_generic_kernel<T>, dim3((unsigned)ceil_div(out_rows, kWavesPerBlock)), dim3(kBlock), 0, st, p, gx);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

}  // namespace pglamd
