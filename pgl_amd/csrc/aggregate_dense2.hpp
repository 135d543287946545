// aggregate_dense2.hpp -- aggregation feeding a dense layer, second form (round 3): the workgroup is SPECIALISED.
//
//   GCNConv: out = act( (dst_scale * sum_{u->v} x[u]) @ W + bias )          (pgl/nn/conv.py:242-254)
//
// Why a second form.  The first one (flat kernel, SINK = 1) lets every wave park its finished rows in a tile of its own and
// multiply the tile by W itself.  Measured at C2 it LOSES to aggregate-then-GEMM (1.65 vs 1.46 ms), and an occupancy experiment
// (profiles/r03/flat_kernel_occupancy_experiment.txt: the plain flat kernel is as fast with 16 resident waves per CU as with
// 32) says it is not the occupancy: it is the B operand.  Every 16-row tile re-reads all of W (64 KB) through the vector-memory
// path the row gathers live on -- +40 % requests -- and the gathering wave stands still while it multiplies.
//
// Here a workgroup is 8 PRODUCER waves + 4 MATRIX waves:
//   * producers run the flat kernel's edge walk (scalar index loads, 8 row gathers in flight twice over) and hand each finished
//     row -- scaled -- to a ring of row slots in LDS (one LDS atomic for the slot, one flag store to publish);
//   * a matrix wave claims a tile of 16 published rows: A operand from the ring (rows padded by 4 floats: conflict-free), B operand from
//     a copy of W that lives in LDS for the whole life of the workgroup (packed in MFMA order, ds_read_b128), v_mfma_f32_16x16x4_f32
//     in fp32 (the reference's arithmetic up to re-association), bias + activation on the result, and only that is stored.
//   W never touches the vector-memory path after the first 64 KB load, the producers never wait for a multiplication, and the
//   [N, d_in] intermediate never leaves the chip (unless the caller keeps it for the weight gradient).
// LDS: W (<= 64 KB) + ring (28 rows) + flags = < 80 KB, two workgroups (24 waves, 16 of them gathering) per CU.
// Rows longer than a chunk keep the partial / fix-up path of the flat kernel; the fix-up leaves the finished rows in their
// partial slots and dense_hub_kernel applies the layer to them 16 at a time.  Rows without edges get act(bias) from a matrix wave.
#pragma once
#include "aggregate_flat.hpp"

namespace pglamd {

constexpr int kD2Prod = 8;                           // producer waves per workgroup
constexpr int kD2Cons = 4;                           // matrix waves per workgroup: waves 0..3, one per SIMD
constexpr int kD2Threads = (kD2Prod + kD2Cons) * kWave;
#ifndef PGLAMD_D2_SLEEP
#define PGLAMD_D2_SLEEP 4
#endif
#ifndef PGLAMD_D2_RING
#define PGLAMD_D2_RING 28
#endif
constexpr int kD2Ring = PGLAMD_D2_RING;                          // row slots (>= 16: a tile must fit; 28 keeps the workgroup under 80 KB)
constexpr int kD2SpinLimit = 1 << 22;                // bound of every LDS wait (~0.3 s of spinning; real waits are microseconds).  A wait that runs into
                                                     // it is a protocol bug: the wave TRAPS -- the launch fails loudly (hipErrorLaunchFailure at the next
                                                     // synchronisation) instead of hanging the GPU or returning rows that were never multiplied.

inline size_t dense2_lds_bytes(int d_in, int d_out) {
    return ((size_t)d_in * d_out + (size_t)kD2Ring * (d_in + 4)) * sizeof(float) + (2 * kD2Ring + 12) * sizeof(int);
}
inline bool dense2_covers(int64_t d_in, int64_t d_out) {
    return (d_in == 64 || d_in == 128) && d_out % 16 == 0 && d_out > 0 && dense2_lds_bytes((int)d_in, (int)d_out) <= 80 * 1024;
}

// rows without any edge: aggregate 0, layer output act(bias)
__global__ __launch_bounds__(kBlock) void dense_empty_rows_kernel(AggParams p) {
    const int lane = threadIdx.x & (kWave - 1);
    if (p.out) zero_empty_rows_role<float>(p, (int64_t)blockIdx.x, lane);
    dense_empty_rows_role(p, (int64_t)blockIdx.x, lane);
}

// One step of the empty-row work inside the fused kernel: rows [64 step, 64 step + 64); those without any edge get act(bias)
// (and a zero aggregate where the caller keeps it).
template <int VEC>
__device__ __forceinline__ void d2_empty_rows_step(const AggParams& p, int step, int lane) {
    using V = VecT<float, VEC>;
    const int64_t r0 = (int64_t)step * kWave, r = r0 + lane;
    bool empty = false;
    if (r < p.out_rows) empty = r >= p.n_csr_rows || p.zero_indptr[r] == p.zero_indptr[r + 1];
    unsigned long long m = __ballot(empty);
    float* keep = static_cast<float*>(p.out);
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        float* dst = p.out2 + (r0 + l) * (int64_t)p.dout2;
        for (int j = lane; j < p.dout2; j += kWave) {
            float v = p.bias ? p.bias[j] : 0.f;
            if (p.act) v = v > 0.f ? v : 0.f;
            dst[j] = v;
        }
        if (keep) *reinterpret_cast<V*>(keep + (r0 + l) * p.ldo + lane * VEC) = V{};
    }
}

// ES: every edge carries a scale, laid out along the sorted stream (p.src_scale[position]; AggParams::ss_by_pos) -- GCN's source-side
// degree norm without a pass over [N, d_in] before the launch: 4 sequential bytes per edge, one vector load per batch of 8 edges.
template <int VEC, bool ES = false>
__global__ __launch_bounds__(kD2Threads, 6) void agg_dense2_kernel(AggParams p) {
    constexpr int U = 8;
    constexpr int DIN = kWave * VEC;
    constexpr int KK = DIN / 4;                        // k-steps of 4
    using V = VecT<float, VEC>;
    extern __shared__ __align__(16) float d2_lds[];
    const int dout = p.dout2;
    float* wl = d2_lds;                                // W in MFMA order
    constexpr int RS = DIN + 4;                        // ring row stride: +4 floats keep the A-operand reads bank-conflict free
    float* ring = d2_lds + (size_t)DIN * dout;         // [kD2Ring][RS]
    int* ring_row = reinterpret_cast<int*>(ring + kD2Ring * RS);
    int* flag = ring_row + kD2Ring;                    // flag[pos] == slot + 1: slot's row is in place
    int* ctl = flag + kD2Ring;                         // [0] slots handed out, [1] slots released, [2] producers finished, [3] tiles claimed, [4] chunks claimed, [5] batches published, [8..11] their first chunks
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = wave_uniform(threadIdx.x >> 6);
    // Persistent workgroups: the grid is what fits on the chip (two per CU).  Each XCD walks one contiguous eighth of the edge
    // stream (neighbouring rows stay in one L2); its workgroups draw BATCHES of 8 chunks from the XCD's counter in memory, so the
    // launch ends within one chunk's time on every CU whatever the rows look like (hub stretches and low-degree stretches differ
    // 10x in rows per chunk).  A device-scope atomic takes microseconds, so no producer ever issues one: a MATRIX wave keeps two
    // batches queued in LDS, and a producer CLAIMS its next chunk from an LDS counter.  (Producers claiming straight from memory,
    // even one chunk ahead: 1.32 -> 1.66 ms at C2 -- the compiler's uniform-atomic lowering waits for the return on the spot.)
    const int xcd = (int)(blockIdx.x % kXcds);
    const int per_xcd = (p.n_chunks + kXcds - 1) / kXcds;
    const int x_base = xcd * per_xcd;
    const int x_lim = x_base + per_xcd < p.n_chunks ? x_base + per_xcd : p.n_chunks;
    if (x_base >= x_lim) {                              // (fewer chunks than XCDs: this workgroup only has its share of the empty rows)
        if (wave == kD2Cons - 1)
            for (int st = (int)blockIdx.x; st < (int)((p.out_rows + kWave - 1) / kWave); st += (int)gridDim.x) d2_empty_rows_step<VEC>(p, st, lane);
        return;
    }
    {
        // W -> LDS in MFMA B-operand order, straight from the layer's row-major [d_in, d_out] weight (coalesced 16-byte reads,
        // scattered LDS writes):   wl[((ct * KK/4 + k4) * 64 + l) * 4 + j] = w[(4 (4 k4 + j) + l / 16) * d_out + ct * 16 + l % 16]
        const float4* __restrict__ w4 = reinterpret_cast<const float4*>(p.w);
        const int q4 = dout >> 2, n4 = DIN * q4;
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += kD2Threads) {
            const float4 v = w4[i];
            const int row = i / q4, col = (i - row * q4) << 2;
            const int k4 = row >> 4, j = (row >> 2) & 3, l = ((row & 3) << 4) | (col & 15), ct = col >> 4;
            float* d = wl + (((ct * (KK / 4) + k4) * kWave + l) << 2) + j;
            d[0] = v.x; d[4] = v.y; d[8] = v.z; d[12] = v.w;
        }
        if (threadIdx.x < kD2Ring) flag[threadIdx.x] = 0;
        if (threadIdx.x < 12) ctl[threadIdx.x] = 0;
    }
    __syncthreads();

    if (wave < kD2Cons) {
        // ---------------------------------------------------------------- the matrix waves (waves 0..3: one per SIMD)
        // Rows arrive per EDGE walked, matrix work arrives per ROW: a workgroup walking low-degree rows emits 10x the rows of one
        // walking hubs.  Four matrix waves claim tiles from a shared counter; they sleep when there is nothing to multiply.
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int n_ct = dout >> 4;
        const float* __restrict__ bias = p.bias;
        float* __restrict__ out2 = p.out2;
        const int relu = p.act;
        const int q = lane >> 4, l16 = lane & 15;
        // Rows WITHOUT any edge aggregate to 0, so their layer output is act(bias): one matrix wave per workgroup writes those
        // rows (64 rows per step, steps dealt round-robin over the workgroups) before it starts multiplying.
        const int n_steps = (int)((p.out_rows + kWave - 1) / kWave);
        const int z_stride = (int)gridDim.x;
        int zstep = wave == kD2Cons - 1 ? (int)blockIdx.x : n_steps;   // ONE matrix wave per workgroup owns them
        auto empty_step = [&]() { d2_empty_rows_step<VEC>(p, zstep, lane); zstep += z_stride; };
        // the same wave keeps the workgroup's queue of chunk batches two ahead of the producers' claims
        int* xcd_ctr = p.long_count + 8 + xcd;                 // zeroed by the launcher
        int published = 0;
        const bool feeder = wave == kD2Cons - 1;
        bool exhausted = false;
        auto refill = [&]() {
            while (feeder && published - __hip_atomic_load(&ctl[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) / kD2Prod < 2) {
                int g = x_lim;                                   // (the XCD's share is handed out: later batches start "past the end")
                if (!exhausted) {
                    int got = 0;
                    if (lane == 0) got = atomicAdd(xcd_ctr, kD2Prod);
                    g = x_base + wave_uniform(got);
                    if (g >= x_lim) { g = x_lim; exhausted = true; }
                }
                if (lane == 0) ctl[8 + (published & 3)] = g;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                ++published;
                if (lane == 0) __hip_atomic_store(&ctl[5], published, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        refill();
        // (that wave writes the empty rows FIRST and multiplies afterwards: interleaving the two made tiles wait behind row writes,
        //  the ring filled up and the producers stalled -- 1.16 -> 1.26 ms)
        while (zstep < n_steps) { empty_step(); refill(); }
        for (;;) {
            refill();
            int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(&ctl[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = wave_uniform(t);
            int n, spins = 0;
            for (;;) {
                const int s = 16 * t + l16;
                const int f = __hip_atomic_load(&flag[s % kD2Ring], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((__ballot(f == s + 1) & 0xFFFFull) == 0xFFFFull) { n = 16; break; }
                if (__hip_atomic_load(&ctl[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == kD2Prod) {
                    const int total = __hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    n = total - 16 * t < 16 ? total - 16 * t : 16;
                    break;
                }
                refill();
                __builtin_amdgcn_s_sleep(PGLAMD_D2_SLEEP);
                if (++spins > kD2SpinLimit) __builtin_trap();
            }
            if (n <= 0) break;
            // A operand: lane l holds row (l % 16) of the tile, columns 4 kk + l / 16
            const int pos = (16 * t + l16) % kD2Ring;
            const float* trow = ring + pos * RS + q;
            float a[KK];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) a[kk] = trow[4 * kk];
            int rid[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rid[i] = ring_row[(16 * t + 4 * q + i) % kD2Ring];
            // the slots are free again once their contents sit in registers; tiles are released in order (the producers compare
            // their slot number with ONE counter), which only serialises these few LDS reads, not the multiplications
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            for (spins = 0; __hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 16 * t; ++spins) {
                if (spins > kD2SpinLimit) __builtin_trap();
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) __hip_atomic_store(&ctl[1], 16 * t + n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef PGLAMD_D2_SKIP_MFMA                       // experiment: the producers alone (tiles are released, nothing is multiplied)
            continue;
#endif
            // two column tiles at a time (two independent MFMA chains); the B operand of k-step group k4 + 1 is fetched from LDS
            // while group k4 multiplies (the scheduling barrier keeps the compiler from hoisting ALL of W's loads to the top)
            const float4* __restrict__ wl4 = reinterpret_cast<const float4*>(wl);
#pragma unroll 1
            for (int ct0 = 0; ct0 < n_ct; ct0 += 2) {
                const int ct1 = ct0 + 1 < n_ct ? ct0 + 1 : ct0;               // odd tile count: the last pair repeats a tile
                const float4* w0 = wl4 + (ct0 * (KK / 4)) * kWave + lane;
                const float4* w1 = wl4 + (ct1 * (KK / 4)) * kWave + lane;
                f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = f4{0.f, 0.f, 0.f, 0.f};
                float4 b0 = w0[0], b1 = w1[0];
#pragma unroll
                for (int k4 = 0; k4 < KK / 4; ++k4) {
                    float4 nb0 = b0, nb1 = b1;
                    if (k4 + 1 < KK / 4) { nb0 = w0[(k4 + 1) * kWave]; nb1 = w1[(k4 + 1) * kWave]; }
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * k4 + 0], b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * k4 + 0], b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * k4 + 1], b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * k4 + 1], b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * k4 + 2], b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * k4 + 2], b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * k4 + 3], b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * k4 + 3], b1.w, acc1, 0, 0, 0);
                    b0 = nb0; b1 = nb1;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && ct1 == ct0) break;
                    const f4 accj = j == 0 ? acc0 : acc1;
                    int lc = l16;
                    asm volatile("" : "+v"(lc));                   // (per-lane column pointers are formed here, not carried through the MFMA loop)
                    const int colj = (ct0 + j) * 16 + lc;
                    const float bv = bias ? bias[colj] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {              // lane l holds C[4 (l / 16) + i][l % 16]
                        if (4 * q + i < n) {
                            float v = accj[i] + bv;
                            if (relu) v = v > 0.f ? v : 0.f;
                            int ri = rid[i];
                            asm volatile("" : "+v"(ri));       // the 64-bit row addresses are formed HERE, not kept across the MFMA loop
                            out2[(int64_t)ri * dout + colj] = v;
                        }
                    }
                }
            }
        }
        return;
    }

    // -------------------------------------------------------------------- producers: the flat kernel's edge walk
    const cptr<int> rowp = as_const(p.row);
    const cptr<int> colp = as_const(p.col);
    const cptr<int64_t> ipc = as_const(p.indptr);
    const float* __restrict__ x = static_cast<const float*>(p.x);
    const cptr<AggParams> kargs = (cptr<AggParams>)__builtin_amdgcn_kernarg_segment_ptr();
    auto cold = [&]() -> cptr<AggParams> {
        cptr<AggParams> qq = kargs;
        asm volatile("" : "+s"(qq));
        return qq;
    };
    const int j0 = lane * VEC;
    float acc[VEC];
    int c = 0, cur = 0;
    bool head_open = false;
    auto reset = [&]() {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    };
    auto store_partial = [&](bool head) {
        const cptr<AggParams> qq = cold();
        float* dst = static_cast<float*>(head ? qq->part_head : qq->part_tail) + (int64_t)c * DIN;
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = acc[k];
        *reinterpret_cast<V*>(dst + j0) = o;
        if (!head && lane == 0) qq->long_list[atomicAdd(qq->long_count, 1)] = c;
    };
    // a finished row (wholly inside this chunk): scale, keep if asked, hand to the matrix wave
    auto park = [&](int r) {
        const cptr<AggParams> qq = cold();
        if (r >= qq->out_rows) return;
        const float* dsp = qq->dst_scale;
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = acc[k];
        if (qq->is_mean) {
            const cptr<int64_t> ipq = as_const(qq->indptr);
            const float n = (float)(ipq[r + 1] - ipq[r]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = o.v[k] / n;
        }
        if (dsp) {
            const float ds = as_const(dsp)[r];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = o.v[k] * ds;
        }
        float* keep = static_cast<float*>(qq->out);
        if (keep) *reinterpret_cast<V*>(keep + (int64_t)r * qq->ldo + j0) = o;
#ifdef PGLAMD_D2_NO_PARK                         // experiment: the edge walk alone (rows are dropped)
        return;
#endif
        int slot = 0;
        if (lane == 0) slot = __hip_atomic_fetch_add(&ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        slot = wave_uniform(slot);
        for (int spins = 0; slot - __hip_atomic_load(&ctl[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= kD2Ring; ++spins) {
            if (spins > kD2SpinLimit) __builtin_trap();
            __builtin_amdgcn_s_sleep(1);
        }
        const int pos = slot % kD2Ring;
        *reinterpret_cast<V*>(ring + pos * RS + j0) = o;
        if (lane == 0) ring_row[pos] = r;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&flag[pos], slot + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto close_row = [&]() {
        if (head_open) store_partial(true); else park(cur);
        head_open = false;
    };
    auto load_idx = [&](int e, int (&cc)[U], int (&rr)[U]) {
#pragma unroll
        for (int i = 0; i < U; ++i) { rr[i] = rowp[e + i]; cc[i] = colp[e + i]; }
    };
    const float* __restrict__ es_v = p.src_scale;
    auto load_rows = [&](int eb, const int (&cc)[U], V (&vx)[U], float& sv) {
        if constexpr (ES) sv = es_v[eb + (lane & (U - 1))];              // lane i < 8: the scale of edge i of the batch
#pragma unroll
        for (int i = 0; i < U; ++i) vx[i] = *reinterpret_cast<const V*>(x + (int64_t)cc[i] * p.ldx + j0);
    };
    auto consume_one = [&](int r, const V& vx, float s) {
        if (r != cur) { close_row(); cur = r; reset(); }
#pragma unroll
        for (int k = 0; k < VEC; ++k) { if constexpr (ES) acc[k] += vx.v[k] * s; else acc[k] += vx.v[k]; }
    };
    auto lane_scale = [&](float sv, int i) -> float {
        if constexpr (ES) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), i));
        else return 1.f;
    };

    for (;;) {
        int claim_v = 0;
        if (lane == 0) claim_v = __hip_atomic_fetch_add(&ctl[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int k = wave_uniform(claim_v), bno = k / kD2Prod;
        for (int spins = 0; __hip_atomic_load(&ctl[5], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= bno; ++spins) {
            if (spins > kD2SpinLimit) __builtin_trap();
            __builtin_amdgcn_s_sleep(1);
        }
        c = wave_uniform(__hip_atomic_load(&ctl[8 + (bno & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) + k % kD2Prod;
        if (c >= x_lim) break;
        const int e0 = chunk_cut(rowp, ipc, c * p.chunk, p.chunk, p.E);
        const int e1 = chunk_cut(rowp, ipc, c * p.chunk + p.chunk, p.chunk, p.E);
        if (e0 >= e1) continue;
        reset();
        cur = rowp[e0];
        head_open = e0 > 0 && rowp[e0 - 1] == cur;
        int e = e0;
        // three batches deep, as the flat kernel: rows of batch g are consumed while the rows of g+1 are in flight and the
        // (scalar) ids of g+2 are being fetched
        const int n_full = (e1 - e0) / U;
        int cA[U], rA[U], cB[U], rB[U];
        V xA[U];
        float svA = 1.f;
        if (n_full > 0) { load_idx(e, cA, rA); load_rows(e, cA, xA, svA); }
        if (n_full > 1) load_idx(e + U, cB, rB);
        for (int g = 0; g < n_full; ++g) {
            int cC[U], rC[U];
            V xB[U];
            float svB = 1.f;
            const bool more = g + 1 < n_full, more2 = g + 2 < n_full;
            if (more) load_rows(e + U, cB, xB, svB);
            if (more2) load_idx(e + 2 * U, cC, rC);
#pragma unroll
            for (int i = 0; i < U; ++i) consume_one(rA[i], xA[i], lane_scale(svA, i));
            if (more) {
                svA = svB;
#pragma unroll
                for (int i = 0; i < U; ++i) { rA[i] = rB[i]; xA[i] = xB[i]; }
            }
            if (more2) {
#pragma unroll
                for (int i = 0; i < U; ++i) { cB[i] = cC[i]; rB[i] = rC[i]; }
            }
            e += U;
        }
        for (; e < e1; ++e) {
            const int r = rowp[e];
            const V vx = *reinterpret_cast<const V*>(x + (int64_t)colp[e] * p.ldx + j0);
            float s = 1.f;
            if constexpr (ES) { int ei = e; asm volatile("" : "+v"(ei)); s = es_v[ei]; }     // (wave-uniform index through the vector path)
            consume_one(r, vx, s);
        }
        const bool tail_open = e1 < p.E && rowp[e1] == cur;
        if (head_open) store_partial(true);
        else if (tail_open) store_partial(false);
        else park(cur);
        head_open = false;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&ctl[2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The layer applied to the SPLIT rows (rows longer than a chunk): the fix-up kernels leave each finished, scaled row in its
// tail-partial slot (part_tail[a], a = the chunk the row starts in; long_list holds the a's); a wave takes 16 of them, A operand
// and W straight from memory (a few thousand rows: W stays in L2), same MFMA tile and epilogue as the matrix waves.
template <int VEC>
__global__ __launch_bounds__(kBlock) void dense_hub_kernel(AggParams p) {
    // One wave = one (tile of 16 split rows, column tile of 16 outputs): n_tiles x (d_out / 16) independent tasks.  The first version
    // gave a wave a whole tile and walked its column tiles one after the other -- eight dependent rounds of 32 loads + 32 MFMAs on
    // a few hundred waves: 65 us at C2, 5 % of GCNConv's forward (profiles/r04/trace_layers_gcn_relu.txt).
    constexpr int DIN = kWave * VEC, KK = DIN / 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & (kWave - 1);
    const int q = lane >> 4, l16 = lane & 15;
    const int n_tasks = p.long_count[0];
    const int dout = p.dout2, n_ct = dout >> 4;
    const float* __restrict__ pt = static_cast<const float*>(p.part_tail);
    const int64_t wave_g = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t n_work = (int64_t)((n_tasks + 15) / 16) * n_ct;
    for (int64_t wk = wave_g; wk < n_work; wk += n_waves) {
        const int t0 = (int)(wk / n_ct) * 16, ct = (int)(wk % n_ct);
        const int n = n_tasks - t0 < 16 ? n_tasks - t0 : 16;
        const float* arow = pt + (int64_t)p.long_list[t0 + (l16 < n ? l16 : 0)] * p.tile_cols + q;
        const float* wc = p.w + (int64_t)q * dout + ct * 16 + l16;          // B operand: W[4 kk + l / 16][16 ct + l % 16]
        float a[KK], b[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) { a[kk] = arow[4 * kk]; b[kk] = wc[(int64_t)4 * kk * dout]; }
        int rid[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ti = 4 * q + i < n ? 4 * q + i : 0;
            rid[i] = p.row[(int64_t)(p.long_list[t0 + ti] + 1) * p.chunk - 1];
        }
        f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = f4{0.f, 0.f, 0.f, 0.f};       // two chains over the even / odd k-steps
#pragma unroll
        for (int kk = 0; kk < KK; kk += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b[kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk + 1], b[kk + 1], acc1, 0, 0, 0);
        }
        const int colj = ct * 16 + l16;
        const float bv = p.bias ? p.bias[colj] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (4 * q + i < n && rid[i] < p.out_rows) {
                float v = acc0[i] + acc1[i] + bv;
                if (p.act) v = v > 0.f ? v : 0.f;
                p.out2[(int64_t)rid[i] * dout + colj] = v;
            }
        }
    }
}

template <int VEC>
static int32_t launch_dense_hub(const AggParams& p, hipStream_t st) {
    const int64_t work = ceil_div(p.n_chunks, 16) * (p.dout2 >> 4);      // (at most one split row starts per chunk; the real count is on the device)
    hipLaunchKernelGGL((dense_hub_kernel<VEC>), dim3((unsigned)std::min<int64_t>(2048, ceil_div(work, kWavesPerBlock))), dim3(kBlock), 0, st, p);
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

template <int VEC, bool ES = false>
static int32_t launch_dense2(AggParams p, hipStream_t st) {
    const int d_in = kWave * VEC;
    const size_t lds = dense2_lds_bytes(d_in, p.dout2);
    static bool attr_set = false;                           // (per instantiation; idempotent, so a race only repeats it)
    if (!attr_set) {
        PGLAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_dense2_kernel<VEC, ES>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr_set = true;
    }
    static const int n_cu = [] {
        int dev = 0; hipDeviceProp_t pr{};
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) return 256;
        return pr.multiProcessorCount;
    }();
    static const bool dbg = [&] {
        if (!getenv("PGLAMD_D2_DEBUG")) return false;
        int nblk = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, agg_dense2_kernel<VEC, ES>, kD2Threads, lds);
        fprintf(stderr, "[pglamd] agg_dense2_kernel<%d>: %zu bytes of LDS, %d threads -> %d resident workgroups per CU (%s), %d CUs\n", VEC, lds, kD2Threads, nblk, hipGetErrorString(e), n_cu);
        return true;
    }();
    (void)dbg;
    // two workgroups per CU stay resident for the whole launch (fewer when there are not enough chunks to go round)
    const int64_t nb = std::min<int64_t>((int64_t)2 * n_cu, xcd_grid(ceil_div(p.n_chunks, (int64_t)kD2Prod)));
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)nb;
    const bool fixups = needs_fixups(p);
    PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 16 * sizeof(int), st));      // [0..1] fix-up lists, [8..15] the XCDs' batch counters
    if (p.n_chunks == 0) {                                   // no edge at all: every row is act(bias)
        const int64_t zb = ceil_div(ceil_div(p.out_rows, kWave), kWavesPerBlock);
        hipLaunchKernelGGL(dense_empty_rows_kernel, dim3((unsigned)zb), dim3(kBlock), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
        return PGLAMD_OK;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool profiling = prof().on.load(std::memory_order_relaxed);
    if (profiling) {
        { std::lock_guard<std::mutex> lk(prof().mu); prof().last_kernel = std::string("agg_dense2_kernel<") + (VEC == 2 ? "2" : "1") + ">"; }
        PGLAMD_HIP_CHECK(hipEventCreate(&e0));
        PGLAMD_HIP_CHECK(hipEventCreate(&e1));
        PGLAMD_HIP_CHECK(hipEventRecord(e0, st));
    }
    hipLaunchKernelGGL((agg_dense2_kernel<VEC, ES>), dim3((unsigned)p.n_grid_chunks), dim3(kD2Threads), lds, st, p);
    PGLAMD_LAUNCH_CHECK();
    if (profiling) {
        PGLAMD_HIP_CHECK(hipEventRecord(e1, st));
        std::lock_guard<std::mutex> lk(prof().mu);
        prof().ev.emplace_back(e0, e1);
    }
    if (fixups) {
        hipLaunchKernelGGL((agg_fixup_kernel<float, VEC, 1, 0, false>), dim3((unsigned)std::min<int64_t>(kFixGridShort, ceil_div(p.n_chunks, kWavesPerBlock))), dim3(kBlock), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
        hipLaunchKernelGGL((agg_fixup_kernel<float, VEC, 1, 0, true>), dim3((unsigned)std::min<int64_t>(kFixGridLong, p.n_chunks)), dim3(kFixWaves * kWave), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
        return launch_dense_hub<VEC>(p, st);
    }
    return PGLAMD_OK;
}

}  // namespace pglamd
