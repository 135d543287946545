// aggregate_group.hpp -- aggregation of rows of 64..128 bytes (d = 17..32 fp32, 33..64 fp16, 9..16 fp64; also 32..64
// bytes where the lane-per-edge kernel does not apply or loses, and 256 bytes of fp64 / int64): several edges per
// wave instruction.
//
// The flat kernel (aggregate_flat.hpp) spends one wave-wide load per gathered row, so a 128-byte row keeps half of the
// lanes idle and, more importantly, a wave has only 16..32 rows in flight: at these widths the rate is set by rows in
// flight x latency (d = 32 and d = 64 fp32 take the same 0.62 ms at C2).  Here the wave is split into S = 64 / G groups
// of G lanes; a group covers one row with 16-, 8- or 4-byte lanes and walks ITS OWN chunk of the dst-sorted edge
// stream, so one load instruction gathers S rows and a wave has 8 S .. 16 S rows in flight (C2, d = 32 fp32: 0.62 -> 0.33..0.37 ms,
// fp16 d = 64: 0.63 -> 0.41, fp64 d = 16: 0.75 -> 0.37).  Everything that was wave-uniform
// in the flat kernel is group-uniform here: row / column ids of a batch are one coalesced vector load per group (lane
// l: edge l) forwarded with ds_bpermute, row-boundary tests and row stores are per-lane predicated.  Chunking, the
// T/H partials of rows longer than a chunk and the fix-up pass are the flat kernel's (chunk = one GROUP's edges).
#pragma once
#include "aggregate.hpp"

#include <algorithm>
#include <type_traits>

namespace pglamd {

int group_wave_edges();      // aggregate.hip: edges one WAVE walks (1024; PGLAMD_GCHUNK)
int group_row_bytes();       // aggregate.hip: widest row taken by this kernel (PGLAMD_GROUP_BYTES, 0 disables)
int group_min_bytes();       // aggregate.hip: sum / mean rows up to this many bytes stay with the lane-per-edge kernel (64; PGLAMD_GROUP_MIN_BYTES)
int group_chunk_edges(int groups_per_wave);

// chunk_cut (common.hpp) through the vector path: the chunk index differs between the groups of a wave
__device__ __forceinline__ int group_cut(const int* __restrict__ rowp, const int64_t* __restrict__ ip, int pos, int K, int E) {
    if (pos <= 0) return 0;
    if (pos >= E) return E;
    const int r = rowp[pos];
    const int64_t rs = ip[r];
    if (rs == pos) return pos;
    const int64_t re = ip[r + 1];
    if (re - rs > K) return pos;
    return (int)re;
}

// min is computed as max over order-reversed values (floats: negated, integers: complemented -- both exact), so the hot loop
// has one v_max per element whatever the op; values are turned back wherever they leave the registers
template <typename A> __device__ __forceinline__ A order_flip(A v, bool neg) {
    if constexpr (std::is_floating_point_v<A>) return neg ? -v : v;
    else return neg ? ~v : v;
}
template <typename A> __device__ __forceinline__ A max_of(A a, A b) {
    if constexpr (std::is_same_v<A, float>) return __builtin_fmaxf(a, b);
    else if constexpr (std::is_same_v<A, double>) return __builtin_fmax(a, b);
    else return a > b ? a : b;
}

// 16 rows per batch while they are cheap to hold (<= 8-byte lanes, >= 16-lane groups), 8 otherwise
template <typename T, int VEC, int G> constexpr int group_batch() { return (G >= 16 && VEC * (int)sizeof(T) <= 8) ? 16 : 8; }

template <typename T, int VEC, int G, int RCLS, bool TWO = false>
__global__ __launch_bounds__(kBlock) void agg_group_kernel(AggParams p) {
    constexpr int S = kWave / G;
    constexpr int UB = group_batch<T, VEC, G>();  // edges per batch of a group
    constexpr int NI = UB > G ? UB / G : 1;       // index registers per batch (lane l of a group holds edges l, l + G, ...)
    using V = VecT<T, VEC>;
    using A = typename AccT<T>::type;
    using VA = VecT<A, VEC>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    if ((int)blockIdx.x >= p.n_grid_chunks) {    // trailing blocks: zero-fill rows that receive no edge
        zero_empty_rows_role<T>(p, (int64_t)blockIdx.x - p.n_grid_chunks, lane);
        return;
    }
    const int64_t lb = xcd_swizzle(blockIdx.x, p.n_blocks);
    if (lb < 0) return;
    const int gl = lane & (G - 1), gbase = lane & ~(G - 1);
    const int c = ((int)lb * kWavesPerBlock + wib) * S + lane / G;      // this group's chunk
    const int* __restrict__ rowp = p.row;
    const int* __restrict__ colp = p.col;
    int e0 = 0, e1 = 0;
    if (c < p.n_chunks) {
        e0 = p.align ? group_cut(rowp, p.indptr, c * p.chunk, p.chunk, p.E) : c * p.chunk;
        e1 = p.align ? group_cut(rowp, p.indptr, c * p.chunk + p.chunk, p.chunk, p.E) : min(c * p.chunk + p.chunk, p.E);
    }
    const bool is_max = p.is_max != 0;
    const int j0 = gl * VEC;
    const bool act = j0 < p.tile_cols;
    const T* __restrict__ x = static_cast<const T*>(p.x) + p.j_base + j0;
    const T* __restrict__ x2 = static_cast<const T*>(p.x2) + p.j_base + j0;     // second source table (see aggregate_flat.hpp)
    const int xs = p.x_split;

    A acc[VEC];
    auto reset = [&]() {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = RCLS == 0 ? A(0) : Limits<A>::lo();
    };
    reset();
    const bool neg = RCLS == 1 && !is_max;
    int cur = -1;
    bool head_open = false;
    if (e0 < e1) {
        cur = rowp[e0];
        head_open = e0 > 0 && rowp[e0 - 1] == cur;       // the row began in an earlier chunk
    }

    // Row stores happen for SOME group at almost every step (8 groups x 1/degree), so what they need stays in SGPRs (the
    // flat kernel, short of SGPRs, re-reads it from the kernarg segment at each store: a scalar-load round trip per row).
    auto store_partial = [&](bool head) {
        const AggParams* q = &p;
        A* dst = static_cast<A*>(head ? q->part_head : q->part_tail) + (int64_t)c * q->tile_cols;
        if (act) {
            VA o;
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = RCLS == 1 ? order_flip(acc[k], neg) : acc[k];
            *reinterpret_cast<VA*>(dst + j0) = o;
        }
        if (!head && gl == 0) q->long_list[atomicAdd(q->long_count, 1)] = c;     // this chunk owns the row's fix-up
    };
    auto store_final = [&](int r) {      // (row wholly inside the chunk: a mean divides by its degree, read from indptr)
        const AggParams* q = &p;
        if (r >= q->out_rows || !act) return;
        T* dst = static_cast<T*>(q->out) + (int64_t)r * q->ldo + q->j_base + j0;
        const float* dsp = q->dst_scale;
        A ov[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) ov[k] = RCLS == 1 ? order_flip(acc[k], neg) : acc[k];
        if constexpr (RCLS == 0) {
            if (q->is_mean != 0) {
                const int64_t* ipq = q->indptr;
                const int64_t n = ipq[r + 1] - ipq[r];
#pragma unroll
                for (int k = 0; k < VEC; ++k) ov[k] = ov[k] / (A)n;
            }
            if constexpr (std::is_floating_point_v<A>) {
                if (dsp) {
                    const A ds = (A)dsp[r];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) ov[k] = ov[k] * ds;
                }
            }
        }
        if (q->accumulate == 1) {
            const V old = *reinterpret_cast<const V*>(dst);
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
        *reinterpret_cast<V*>(dst) = o;
    };
    auto consume = [&](int r, const V& vx) {
        if (r != cur) {
            if (head_open) store_partial(true); else store_final(cur);
            head_open = false;
            cur = r; reset();
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const A m = to_acc<T>(vx.v[k]);
            if constexpr (RCLS == 0) acc[k] += m;
            else acc[k] = max_of(acc[k], order_flip(m, neg));
        }
    };

    // three stages, as in the flat kernel: ids of batch g+2 are issued before the rows of batch g+1 (vector loads retire in
    // issue order), rows of g+1 are in flight while batch g is reduced.  Group trip counts differ; finished groups idle.
    auto load_iv = [&](int eb, int (&rv)[NI], int (&cv)[NI]) {
        if (eb < e1) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int pos = min(eb + k * G + (gl & (UB - 1)), e1 - 1);
                rv[k] = rowp[pos];
                cv[k] = colp ? colp[pos] : pos;
            }
        }
    };
    auto load_rows = [&](int eb, const int (&cv)[NI], V (&vx)[UB]) {
        if (eb < e1) {
#pragma unroll
            for (int i = 0; i < UB; ++i) {
                const int cc = __shfl(cv[i / G], gbase + (i % G), kWave);
                if (act && eb + i < e1) vx[i] = *reinterpret_cast<const V*>((TWO ? (cc < xs ? x : x2) : x) + (int64_t)cc * p.ldx);
            }
        }
    };
    int e = e0;
    int rvA[NI] = {}, cvA[NI] = {}, rvB[NI] = {}, cvB[NI] = {};
    V xA[UB];
    load_iv(e, rvA, cvA);
    load_iv(e + UB, rvB, cvB);
    load_rows(e, cvA, xA);
    while (e < e1) {
        int rvC[NI] = {}, cvC[NI] = {};
        V xB[UB];
        load_iv(e + 2 * UB, rvC, cvC);
        load_rows(e + UB, cvB, xB);
        const int nb = min(UB, e1 - e);
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int r = __shfl(rvA[i / G], gbase + (i % G), kWave);
            if (i < nb) consume(r, xA[i]);
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) { rvA[k] = rvB[k]; cvA[k] = cvB[k]; rvB[k] = rvC[k]; cvB[k] = cvC[k]; }
#pragma unroll
        for (int i = 0; i < UB; ++i) xA[i] = xB[i];
        e += UB;
    }
    if (e0 < e1) {       // the row open at the end of the chunk
        const bool tail_open = e1 < p.E && rowp[e1] == cur;
        if (head_open) store_partial(true);                 // middle or closing piece of a long row
        else if (tail_open) store_partial(false);           // first piece of a long row that continues
        else store_final(cur);
    }
}

template <typename T, int VEC, int G, int RCLS>
int32_t launch_group_one(AggParams p, hipStream_t st) {
    constexpr int S = kWave / G;
    const int64_t nb = ceil_div(p.n_chunks, (int64_t)kWavesPerBlock * S);
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    const int64_t zb = p.accumulate ? 0 : ceil_div(ceil_div(p.out_rows, kWave), kWavesPerBlock);
    if (needs_fixups(p)) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool profiling = prof().on.load(std::memory_order_relaxed);
    if (profiling) {
        char name[96];
        snprintf(name, sizeof(name), "agg_group_kernel<%d-byte elements, %d, %d, %d>", (int)sizeof(T), VEC, G, RCLS);
        { std::lock_guard<std::mutex> lk(prof().mu); prof().last_kernel = name; }
        PGLAMD_HIP_CHECK(hipEventCreate(&ev0));
        PGLAMD_HIP_CHECK(hipEventCreate(&ev1));
        PGLAMD_HIP_CHECK(hipEventRecord(ev0, st));
    }
    if (p.x_split != INT32_MAX) hipLaunchKernelGGL((agg_group_kernel<T, VEC, G, RCLS, true>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((agg_group_kernel<T, VEC, G, RCLS, false>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
    PGLAMD_LAUNCH_CHECK();
    if (profiling) {
        PGLAMD_HIP_CHECK(hipEventRecord(ev1, st));
        std::lock_guard<std::mutex> lk(prof().mu);
        prof().ev.emplace_back(ev0, ev1);
    }
    return PGLAMD_OK;
}

// A group spans CB = 64, 128 or 256 bytes of row: G = CB / (lane bytes), 4 <= G <= 32.
template <typename T, int VEC, int CB>
int32_t launch_group_vec(AggParams p, int rcls, int32_t dtype, char* ws, size_t ws_bytes, hipStream_t st, bool* handled) {
    constexpr int G = CB / (VEC * (int)sizeof(T));
    if constexpr (G >= 4 && G <= 32) {
        if (p.tile_cols > G * VEC || p.tile_cols > kWave) return PGLAMD_OK;     // (fix-up: one column per lane)
        using A = typename AccT<T>::type;
        p.chunk = group_chunk_edges(kWave / G);
        p.n_chunks = (int)ceil_div(p.E, p.chunk);
        const size_t half = align_up((size_t)p.n_chunks * p.tile_cols * sizeof(A), 256);
        const size_t lst = align_up((size_t)(p.n_chunks + 64) * sizeof(int), 256);
        if (!ws || ws_bytes < 2 * half + 2 * lst) return PGLAMD_OK;       // workspace sized for another chunking: flat path
        p.part_head = ws;
        p.part_tail = ws + half;
        p.long_count = reinterpret_cast<int*>(ws + 2 * half);
        p.long_list = p.long_count + 64;
        p.long_list2 = reinterpret_cast<int*>(ws + 2 * half + lst);
        *handled = true;
        const int32_t rc = rcls == 0 ? launch_group_one<T, VEC, G, 0>(p, st) : launch_group_one<T, VEC, G, 1>(p, st);
        if (rc != PGLAMD_OK || !needs_fixups(p)) return rc;
        return launch_fixup_cols(p, dtype, rcls, st);
    }
    return PGLAMD_OK;
}

template <typename T, int CB>
int32_t launch_group_class(const AggParams& p, int vec, int rcls, int32_t dtype, char* w, size_t ws_bytes, hipStream_t st, bool* handled) {
    if (vec == 1) return launch_group_vec<T, 1, CB>(p, rcls, dtype, w, ws_bytes, st, handled);
    if (vec == 2) return launch_group_vec<T, 2, CB>(p, rcls, dtype, w, ws_bytes, st, handled);
    if constexpr (sizeof(T) <= 4) {
        if (vec == 4) return launch_group_vec<T, 4, CB>(p, rcls, dtype, w, ws_bytes, st, handled);
    }
    if constexpr (sizeof(T) == 2) {
        if (vec == 8) return launch_group_vec<T, 8, CB>(p, rcls, dtype, w, ws_bytes, st, handled);
    }
    return PGLAMD_OK;
}

// vec: widest lane (elements) the row length and the pointers allow
template <typename T>
int32_t launch_group(const AggParams& p, int vec, int rcls, int32_t dtype, void* ws, size_t ws_bytes, hipStream_t st, bool* handled) {
    *handled = false;
    char* w = static_cast<char*>(ws);
    const size_t rb = (size_t)p.tile_cols * sizeof(T);
    if (rb <= 64) return launch_group_class<T, 64>(p, vec, rcls, dtype, w, ws_bytes, st, handled);
    if (rb <= 128) return launch_group_class<T, 128>(p, vec, rcls, dtype, w, ws_bytes, st, handled);
    // 256-byte rows are byte-bound in the flat kernel already (d = 64 fp32: 0.63 ms either way); only the 8-byte types,
    // which the flat kernel covers with 8-byte lanes, gain here (fp64 d = 32: 0.79 -> 0.63 ms)
    if constexpr (sizeof(T) == 8) {
        if (rb <= 256) return launch_group_class<T, 256>(p, vec, rcls, dtype, w, ws_bytes, st, handled);
    }
    return PGLAMD_OK;
}

}  // namespace pglamd
