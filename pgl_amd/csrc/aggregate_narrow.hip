// aggregate_narrow.hip -- K1/K2/K5 for NARROW rows (<= 16 elements, <= 64 bytes of accumulator): one LANE per edge.
//
//   out[r, 0:d] = dst_scale[r] * REDUCE_{p: row[p]==r} ( src_scale[col[p]] * x[col[p], 0:d] (mop) y[yp, :] )
//
// Same contract, chunking, partial layout and fix-up as aggregate.hip (pgl/graph.py:859-861, 885-887, 929-937,
// pgl/math.py:30-178); what differs is the lane geometry.  With lanes across the feature dimension a [E,8] operand
// (attention logits per head, pgl/nn/conv.py:331-339) keeps 8 of 64 lanes busy and a [E] operand one.  Here:
//   * a wave still owns one chunk of the dst-sorted edge stream (chunk_cut: only rows longer than a chunk split),
//     but walks it 64 edges at a time, lane l holding edge eb+l: its row id, its column id and its whole
//     gathered operand row (d values in registers, one 4..64-byte load per lane, 4 batches = 256 gathers in flight);
//   * rows are contiguous runs of lanes.  A ballot of the run heads gives every lane the first lane of its run,
//     a 6-step Hillis-Steele scan restricted to the run (ds_bpermute) leaves the run's reduction in its last lane,
//     a wave-uniform carry joins runs across batches;
//   * the lane that holds the last edge of a row stores it (consecutive rows -> consecutive lanes -> dense stores);
//     the pieces of split rows go to the same T[a] / H[c] partial arrays and task list as in aggregate.hip and are
//     combined by the same fix-up kernels, in chunk order => bit-reproducible, no atomics on data.
#include "aggregate.hpp"

#include <algorithm>

namespace pglamd {

namespace {

template <typename T> __device__ __forceinline__ T shfl_up_t(T v, int off) { return __shfl_up(v, off, kWave); }

// value of lane (l - OFF) within the lane's 16-lane DPP row (row_shr:OFF; lanes whose source falls outside the row keep
// their own value -- the caller masks them) and of one wave-uniform lane, for 32- and 64-bit element types: VALU / SALU
// moves, no LDS round trip.
template <int OFF> __device__ __forceinline__ unsigned dpp_row_shr_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x110 + OFF, 0xF, 0xF, false);
}
template <int OFF, typename T> __device__ __forceinline__ T dpp_row_shr(T v) {
    if constexpr (sizeof(T) == 4) {
        unsigned u; __builtin_memcpy(&u, &v, 4);
        u = dpp_row_shr_u32<OFF>(u);
        T o; __builtin_memcpy(&o, &u, 4);
        return o;
    } else {
        unsigned u[2]; __builtin_memcpy(u, &v, 8);
        u[0] = dpp_row_shr_u32<OFF>(u[0]); u[1] = dpp_row_shr_u32<OFF>(u[1]);
        T o; __builtin_memcpy(&o, u, 8);
        return o;
    }
}
template <typename T> __device__ __forceinline__ T read_lane(T v, int src_lane) {       // src_lane wave-uniform
    if constexpr (sizeof(T) == 4) {
        unsigned u; __builtin_memcpy(&u, &v, 4);
        u = (unsigned)__builtin_amdgcn_readlane((int)u, src_lane);
        T o; __builtin_memcpy(&o, &u, 4);
        return o;
    } else {
        unsigned u[2]; __builtin_memcpy(u, &v, 8);
        u[0] = (unsigned)__builtin_amdgcn_readlane((int)u[0], src_lane); u[1] = (unsigned)__builtin_amdgcn_readlane((int)u[1], src_lane);
        T o; __builtin_memcpy(&o, u, 8);
        return o;
    }
}
template <typename T> __device__ __forceinline__ T shfl_t(T v, int src) { return __shfl(v, src, kWave); }

// Rescale factor exp(v), v <= 0, between two running maxima.  fp32 uses the hardware exp2 (relative error
// ~|v| * 2^-24: < 1e-6 wherever the factor is large enough to matter, i.e. |v| < 16).
template <typename A> __device__ __forceinline__ A exp_a(A v);
template <> __device__ __forceinline__ float exp_a<float>(float v) { return __expf(v); }
template <> __device__ __forceinline__ double exp_a<double>(double v) { return exp(v); }

// (m, s) = running maximum and sum of exp(x - m): the pair a softmax needs per segment and column.
// Merging two pairs rescales the one with the smaller maximum; (-inf, 0) is the identity.
template <typename A> __device__ __forceinline__ void softmax_merge(A m1, A s1, A m2, A s2, A& m, A& s) {
    if (m1 >= m2) { const A f = m1 == m2 ? A(1) : exp_a<A>(m2 - m1); m = m1; s = s1 + s2 * f; }
    else { const A f = exp_a<A>(m1 - m2); m = m2; s = s1 * f + s2; }
}

// RCLS: 0 sum / mean, 1 max / min, 2 softmax statistics (out[r] = 2*d values: the segment maxima, then the segment sums
//       of exp(x - max), one 64-byte line for d = 8 fp32; partials use the same layout).   YMODE: 0 none, 1 y is [E] / [E,1] (one value per edge), 2 y is [E,d]
template <typename T, int D, int RCLS, int YMODE, bool TWO = false>
__global__ __launch_bounds__(kBlock) void agg_narrow_kernel(AggParams p) {
    constexpr int NB = D * sizeof(typename AccT<T>::type) >= 64 ? 2 : 4;   // batches of 64 edges whose loads are issued together (fewer for 64-byte rows: registers)
    constexpr int VL = (D * sizeof(T) >= 16) ? (int)(16 / sizeof(T)) : D;   // elements per load instruction
    using A = typename AccT<T>::type;
    using VLoad = VecT<T, VL>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    const int d = p.tile_cols;
    const bool exact = p.narrow_vec != 0 && d % VL == 0;    // rows move as whole VL-element vectors (d = 12: three 16-byte loads)
    const bool is_max = p.is_max != 0;
    T* __restrict__ out = static_cast<T*>(p.out);

    if ((int)blockIdx.x >= p.n_grid_chunks) {              // trailing blocks: rows without edges get 0, one lane per row
        if (p.accumulate || RCLS == 2) return;
        const int64_t r = ((int64_t)blockIdx.x - p.n_grid_chunks) * kBlock + threadIdx.x;
        if (r >= p.out_rows) return;
        if (r < p.n_csr_rows && p.zero_indptr[r] != p.zero_indptr[r + 1]) return;
        T* dst = out + r * p.ldo;
        if (exact) {
#pragma unroll
            for (int k0 = 0; k0 < D; k0 += VL)
                if (k0 < d) *reinterpret_cast<VLoad*>(dst + k0) = VLoad{};
        } else {
            for (int k = 0; k < d; ++k) dst[k] = from_acc<T>(A(0));
        }
        return;
    }
    const int64_t lb = xcd_swizzle(blockIdx.x, p.n_blocks);
    if (lb < 0) return;
    const int c = wave_uniform((int)lb * kWavesPerBlock + wib);
    if (c >= p.n_chunks) return;
    const cptr<int> rowc = as_const(p.row);
    const int e0 = p.align ? chunk_cut(rowc, as_const(p.indptr), c * p.chunk, p.chunk, p.E) : c * p.chunk;
    const int e1 = p.align ? chunk_cut(rowc, as_const(p.indptr), c * p.chunk + p.chunk, p.chunk, p.E) : min(c * p.chunk + p.chunk, p.E);
    if (e0 >= e1) return;
    const int first_row = rowc[e0];
    const bool head_open = e0 > 0 && rowc[e0 - 1] == first_row;    // the chunk's first row began in an earlier chunk

    const int* __restrict__ rowp = p.row;
    const int* __restrict__ colp = p.col;
    const int* __restrict__ eidp = p.eid;
    const T* __restrict__ x = static_cast<const T*>(p.x);
    const T* __restrict__ x2 = static_cast<const T*>(p.x2);     // second source table (see aggregate_flat.hpp)
    const int xs = p.x_split;
    const T* __restrict__ y = static_cast<const T*>(p.y);
    const float* __restrict__ sscale = p.src_scale;

    constexpr bool SM = RCLS == 2;
    // min runs as max over order-reversed values (-x for floats, ~x for integers: both exact), so the scan has ONE
    // compare-select per step; values are flipped when loaded and flipped back wherever they leave the kernel
    const bool flip_min = RCLS == 1 && !is_max;
    auto flip = [&](A a) -> A {
        if constexpr (std::is_floating_point_v<A>) return -a; else return ~a;
    };
    auto ident = [&]() -> A { return RCLS == 0 ? A(0) : Limits<A>::lo(); };
    auto comb = [&](A a, A b) -> A {                       // a = earlier edges, b = later edges
        if constexpr (RCLS == 0) return a + b;
        else if constexpr (std::is_same_v<A, float>) return __builtin_fmaxf(a, b);       // v_max_f32
        else if constexpr (std::is_same_v<A, double>) return __builtin_fmax(a, b);
        else return b > a ? b : a;
    };

    A carry[D], carry_s[SM ? D : 1];
#pragma unroll
    for (int k = 0; k < D; ++k) { carry[k] = ident(); if constexpr (SM) carry_s[k] = A(0); }
    int carry_row = -1;

    for (int eb = e0; eb < e1; eb += kWave * NB) {
        int r[NB], rn[NB], cc[NB], yy[NB];
        bool valid[NB];
        T raw[NB][D];
        T yraw[NB][YMODE == 2 ? D : 1];
        float ss[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int e = eb + b * kWave + lane;
            valid[b] = e < e1;
            r[b] = valid[b] ? rowp[e] : -1;
            rn[b] = (valid[b] && e + 1 < p.E) ? rowp[e + 1] : -1;
            cc[b] = valid[b] ? (colp ? colp[e] : e) : 0;
            if constexpr (YMODE != 0) yy[b] = valid[b] ? (eidp ? eidp[e] : e) : 0;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const T* xr = (TWO ? (cc[b] < xs ? x : x2) : x) + (int64_t)cc[b] * p.ldx;
            if (valid[b]) {
                if (exact) {
#pragma unroll
                    for (int k0 = 0; k0 < D; k0 += VL) {
                        VLoad v{};
                        if (k0 < d) v = *reinterpret_cast<const VLoad*>(xr + k0);
#pragma unroll
                        for (int k = 0; k < VL; ++k) raw[b][k0 + k] = v.v[k];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < D; ++k) raw[b][k] = k < d ? xr[k] : T{};
                }
                if constexpr (YMODE == 1) yraw[b][0] = y[(int64_t)yy[b] * p.ldy];
                if constexpr (YMODE == 2) {
                    const T* yr = y + (int64_t)yy[b] * p.ldy;
                    if (exact) {
#pragma unroll
                        for (int k0 = 0; k0 < D; k0 += VL) {
                            VLoad v{};
                            if (k0 < d) v = *reinterpret_cast<const VLoad*>(yr + k0);
#pragma unroll
                            for (int k = 0; k < VL; ++k) yraw[b][k0 + k] = v.v[k];
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < D; ++k) yraw[b][k] = k < d ? yr[k] : T{};
                    }
                }
                ss[b] = sscale ? sscale[cc[b]] : 1.f;
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ebb = eb + b * kWave;
            if (ebb >= e1) break;                          // wave-uniform
            A v[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                A m = valid[b] ? to_acc<T>(raw[b][k]) : ident();
                if constexpr (std::is_floating_point_v<A>) { if (sscale) m = m * (A)ss[b]; }
                if constexpr (YMODE == 1) { if (valid[b]) m = apply_mop(m, to_acc<T>(yraw[b][0]), p.mop); }
                if constexpr (YMODE == 2) { if (valid[b]) m = apply_mop(m, to_acc<T>(yraw[b][k]), p.mop); }
                if constexpr (RCLS == 1) { if (flip_min && valid[b]) m = flip(m); }
                v[k] = m;
            }
            A sv[SM ? D : 1];
            if constexpr (SM) {
#pragma unroll
                for (int k = 0; k < D; ++k) sv[k] = valid[b] ? A(1) : A(0);
            }
            // run structure of this batch: head lanes, and for every lane the first lane of its run
            const int rp = __shfl_up(r[b], 1, kWave);
            const bool head = lane == 0 || rp != r[b] || !valid[b];
            const unsigned long long hm = __ballot(head);
            const int start = 63 - __builtin_clzll(hm & (~0ull >> (63 - lane)));
            // Segmented inclusive scan.  Phase 1: Hillis-Steele inside each 16-lane DPP row (row_shr 1, 2, 4, 8).  Phase 2: a run
            // that began in an earlier row takes that row's (by then complete) last value, rows 1..3 in order.
            auto merge_in = [&](bool take, const A (&tv)[D], const A (&ts)[SM ? D : 1]) {
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    if constexpr (SM) { if (take) softmax_merge(tv[k], ts[k], v[k], sv[k], v[k], sv[k]); }
                    else v[k] = comb(take ? tv[k] : ident(), v[k]);       // identity for masked lanes: one select + one op
                }
            };
            auto row_step = [&](auto off_tag) {
                constexpr int OFF = decltype(off_tag)::value;
                const bool take = (lane & 15) >= OFF && lane - OFF >= start;
                A tv[D], ts[SM ? D : 1];
#pragma unroll
                for (int k = 0; k < D; ++k) { tv[k] = dpp_row_shr<OFF>(v[k]); if constexpr (SM) ts[k] = dpp_row_shr<OFF>(sv[k]); }
                merge_in(take, tv, ts);
            };
            if constexpr (SM) {
                // Two phases instead of an online-softmax scan (round 2; the scan paid an exponential and two ds_bpermutes per
                // element per step, six steps -- the sequential case was compute-bound at ~0.5 ms for [20 M, 8]):
                //   1. segmented MAX scan over the run (DPP row steps + 3 cross-row reads: VALU only), the run's maximum is
                //      fetched from the run's LAST lane (one ds_bpermute per column);
                //   2. p = exp(x - run max): ONE exponential per element, the accurate one -- these are the values the
                //      normalisation divides by; segmented SUM scan of p (DPP again).
                // Every lane then holds (run max, inclusive prefix sum relative to it): a valid softmax state for the carry /
                // partial merges below, which keep the rescaling form.
                auto seg_scan = [&](A (&val)[D], auto op, A idv) {
                    auto step = [&](auto off_tag) {
                        constexpr int OFF = decltype(off_tag)::value;
                        const bool take = (lane & 15) >= OFF && lane - OFF >= start;
#pragma unroll
                        for (int k = 0; k < D; ++k) { const A t = dpp_row_shr<OFF>(val[k]); val[k] = op(take ? t : idv, val[k]); }
                    };
                    step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
                    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 8>{});
#pragma unroll
                    for (int rw = 1; rw < 4; ++rw) {
                        const bool take = (lane >> 4) == rw && start < 16 * rw;
#pragma unroll
                        for (int k = 0; k < D; ++k) { const A t = read_lane(val[k], 16 * rw - 1); val[k] = op(take ? t : idv, val[k]); }
                    }
                };
                seg_scan(v, [](A a, A b) { return b > a ? b : a; }, Limits<A>::lo());
                const unsigned long long above = lane == kWave - 1 ? 0ull : (hm >> (lane + 1));
                const int tail = above ? lane + __builtin_ctzll(above) : kWave - 1;       // last lane of this lane's run
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const A xk = valid[b] ? to_acc<T>(raw[b][k]) : A(0);
                    const A mk = shfl_t(v[k], tail);
                    v[k] = mk;
                    if constexpr (std::is_same_v<A, float>) sv[k] = valid[b] ? expf(xk - mk) : 0.f;
                    else sv[k] = valid[b] ? exp(xk - mk) : A(0);
                }
                seg_scan(sv, [](A a, A b) { return a + b; }, A(0));
            } else {
            row_step(std::integral_constant<int, 1>{});
            row_step(std::integral_constant<int, 2>{});
            row_step(std::integral_constant<int, 4>{});
            row_step(std::integral_constant<int, 8>{});
            }
#pragma unroll
            for (int rw = 1; rw < (SM ? 1 : 4); ++rw) {
                const bool take = (lane >> 4) == rw && start < 16 * rw;
                A tv[D], ts[SM ? D : 1];
#pragma unroll
                for (int k = 0; k < D; ++k) { tv[k] = read_lane(v[k], 16 * rw - 1); if constexpr (SM) ts[k] = read_lane(sv[k], 16 * rw - 1); }
                merge_in(take, tv, ts);
            }
            if (start == 0 && r[b] == carry_row) {          // run continues the last run of the previous batch
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    if constexpr (SM) softmax_merge(carry[k], carry_s[k], v[k], sv[k], v[k], sv[k]);
                    else v[k] = comb(carry[k], v[k]);
                }
            }
            const int last = min(kWave, e1 - ebb) - 1;      // last valid lane (wave-uniform)
            carry_row = __builtin_amdgcn_readlane(r[b], last);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                carry[k] = read_lane(v[k], last);
                if constexpr (SM) carry_s[k] = read_lane(sv[k], last);
            }

            if (!valid[b]) continue;
            const int e = ebb + lane;
            const bool row_ends = rn[b] != r[b];
            const bool chunk_ends = e == e1 - 1;
            if (!row_ends && !chunk_ends) continue;
            const bool piece_of_earlier = head_open && r[b] == first_row;
            if (piece_of_earlier || !row_ends) {
                // a piece of a split row: H[c] when the row began in an earlier chunk, T[c] when it begins here
                A* dst = static_cast<A*>(piece_of_earlier ? p.part_head : p.part_tail) + (int64_t)c * d * (SM ? 2 : 1);
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k < d) { dst[k] = (RCLS == 1 && flip_min) ? flip(v[k]) : v[k]; if constexpr (SM) dst[d + k] = sv[k]; }
                if (!piece_of_earlier) p.long_list[atomicAdd(p.long_count, 1)] = c;
                continue;
            }
            const int rr = r[b];
            if (rr >= p.out_rows) continue;
            T* dst = out + (int64_t)rr * p.ldo;
            if constexpr (SM) {
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k < d) { dst[k] = from_acc<T>(v[k]); dst[d + k] = from_acc<T>(sv[k]); }
                continue;
            }
            if constexpr (RCLS == 0) {
                if (p.is_mean) {
                    const A n = (A)(p.indptr[rr + 1] - p.indptr[rr]);
#pragma unroll
                    for (int k = 0; k < D; ++k) v[k] = v[k] / n;
                }
                if constexpr (std::is_floating_point_v<A>) {
                    if (p.dst_scale) {
                        const A ds = (A)p.dst_scale[rr];
#pragma unroll
                        for (int k = 0; k < D; ++k) v[k] = v[k] * ds;
                    }
                }
            }
            if constexpr (RCLS == 1) {
                if (flip_min) {
#pragma unroll
                    for (int k = 0; k < D; ++k) v[k] = flip(v[k]);
                }
            }
            if (p.accumulate == 1) {
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k < d) {
                        const A old = to_acc<T>(dst[k]);
                        if constexpr (RCLS == 1) v[k] = is_max ? (v[k] > old ? v[k] : old) : (v[k] < old ? v[k] : old);
                        else v[k] = old + v[k];
                    }
            }
            if (exact) {
#pragma unroll
                for (int k0 = 0; k0 < D; k0 += VL) {
                    VLoad o;
#pragma unroll
                    for (int k = 0; k < VL; ++k) o.v[k] = from_acc<T>(v[k0 + k]);
                    if (k0 < d) *reinterpret_cast<VLoad*>(dst + k0) = o;
                }
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k < d) dst[k] = from_acc<T>(v[k]);
            }
        }
    }
}

// Fix-up of the softmax statistics of split rows: one wave per listed task; lane i merges the partials
// i, i+64, ... of the row's sequence T[a], H[a+1], ..., H[b] in order, then the lanes merge in a fixed tree.
template <typename A, int D>
__global__ __launch_bounds__(kBlock) void softmax_fixup_kernel(AggParams p) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    const int d = p.tile_cols;
    const cptr<int> rowc = as_const(p.row);
    const cptr<int64_t> ip = as_const(p.indptr);
    const A* __restrict__ ph = static_cast<const A*>(p.part_head);
    const A* __restrict__ pt = static_cast<const A*>(p.part_tail);
    const int n_tasks = p.long_count[0];
    for (int t_id = (int)blockIdx.x * kWavesPerBlock + wib; t_id < n_tasks; t_id += (int)gridDim.x * kWavesPerBlock) {
        const int a = wave_uniform(p.long_list[t_id]);
        const int r = rowc[(a + 1) * p.chunk - 1];
        const int b = (int)((ip[r + 1] - 1) / p.chunk);
        A m[D], s[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { m[k] = Limits<A>::lo(); s[k] = A(0); }
        for (int c = a + lane; c <= b; c += kWave) {
            const A* src = (c == a ? pt : ph) + (int64_t)c * d * 2;
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (k < d) softmax_merge(m[k], s[k], src[k], src[d + k], m[k], s[k]);
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const A m2 = __shfl_down(m[k], off, kWave), s2 = __shfl_down(s[k], off, kWave);
                softmax_merge(m[k], s[k], m2, s2, m[k], s[k]);
            }
        }
        if (lane == 0 && r < p.out_rows) {
            A* o = static_cast<A*>(p.out) + (int64_t)r * p.ldo;
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (k < d) { o[k] = m[k]; o[d + k] = s[k]; }
        }
    }
}

template <typename T, int D>
int32_t launch_softmax_stats(AggParams p, hipStream_t st) {
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    if (p.n_chunks > 1) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL((agg_narrow_kernel<T, D, 2, 0>), dim3((unsigned)p.n_grid_chunks), dim3(kBlock), 0, st, p);
    PGLAMD_LAUNCH_CHECK();
    if (p.n_chunks > 1) {
        hipLaunchKernelGGL((softmax_fixup_kernel<T, D>), dim3((unsigned)std::min<int64_t>(1024, nb)), dim3(kBlock), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
    }
    return PGLAMD_OK;
}

template <typename T>
int32_t softmax_stats_typed(const AggParams& p, hipStream_t st) {
    const int d = p.tile_cols;
    if (d <= 1) return launch_softmax_stats<T, 1>(p, st);
    if (d <= 2) return launch_softmax_stats<T, 2>(p, st);
    if (d <= 4) return launch_softmax_stats<T, 4>(p, st);
    if (d <= 8) return launch_softmax_stats<T, 8>(p, st);
    if constexpr (sizeof(T) <= 4) {
        if (d <= 16) return launch_softmax_stats<T, 16>(p, st);
    }
    return fail(PGLAMD_E_SHAPE, "softmax statistics: d=%d beyond the narrow kernel", d);
}

template <typename T, int D, int RCLS, int YMODE>
int32_t launch_one(AggParams p, int32_t dtype, hipStream_t st) {
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    const int64_t zb = p.accumulate ? 0 : ceil_div(p.out_rows, kBlock);
    if (needs_fixups(p)) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool profiling = prof().on.load(std::memory_order_relaxed);
    if (profiling) {
        char name[96];
        snprintf(name, sizeof(name), "agg_narrow_kernel<%d-byte elements, %d, %d, %d>", (int)sizeof(T), D, RCLS, YMODE);
        { std::lock_guard<std::mutex> lk(prof().mu); prof().last_kernel = name; }
        PGLAMD_HIP_CHECK(hipEventCreate(&ev0));
        PGLAMD_HIP_CHECK(hipEventCreate(&ev1));
        PGLAMD_HIP_CHECK(hipEventRecord(ev0, st));
    }
    if (p.x_split != INT32_MAX) hipLaunchKernelGGL((agg_narrow_kernel<T, D, RCLS, YMODE, true>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
    else hipLaunchKernelGGL((agg_narrow_kernel<T, D, RCLS, YMODE, false>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
    PGLAMD_LAUNCH_CHECK();
    if (profiling) {
        PGLAMD_HIP_CHECK(hipEventRecord(ev1, st));
        std::lock_guard<std::mutex> lk(prof().mu);
        prof().ev.emplace_back(ev0, ev1);
    }
    if (needs_fixups(p)) return launch_fixup_cols(p, dtype, RCLS, st);
    return PGLAMD_OK;
}

template <typename T, int D>
int32_t pick_mode(const AggParams& p, int32_t dtype, int rcls, int ymode, hipStream_t st, bool* handled) {
    *handled = true;
    if (rcls == 0) {
        if (ymode == 0) return launch_one<T, D, 0, 0>(p, dtype, st);
        if constexpr (std::is_floating_point_v<T>) {
            if (ymode == 1) return launch_one<T, D, 0, 1>(p, dtype, st);
            if (ymode == 2) return launch_one<T, D, 0, 2>(p, dtype, st);
        }
    } else if (ymode == 0) {
        return launch_one<T, D, 1, 0>(p, dtype, st);
    }
    *handled = false;
    return PGLAMD_OK;
}

template <typename T>
int32_t pick_width(const AggParams& p, int32_t dtype, int rcls, int ymode, hipStream_t st, bool* handled) {
    using A = typename AccT<T>::type;
    const int d = p.tile_cols;
    *handled = false;
    if (d <= 1) return pick_mode<T, 1>(p, dtype, rcls, ymode, st, handled);
    if (d <= 2) return pick_mode<T, 2>(p, dtype, rcls, ymode, st, handled);
    if (d <= 4) return pick_mode<T, 4>(p, dtype, rcls, ymode, st, handled);
    if (d <= 8) return pick_mode<T, 8>(p, dtype, rcls, ymode, st, handled);
    if constexpr (sizeof(A) <= 4) {
        if (d <= 16) return pick_mode<T, 16>(p, dtype, rcls, ymode, st, handled);
    }
    return PGLAMD_OK;
}

}  // namespace

bool narrow_softmax_covers(int64_t d, int32_t dtype) {
    return (dtype == PGLAMD_F32 && d <= 16) || (dtype == PGLAMD_F64 && d <= 8);
}

size_t narrow_softmax_workspace_bytes(int64_t num_rows, int64_t d, int32_t dtype, int chunk) {
    const int64_t n_chunks = ceil_div(num_rows > 0 ? num_rows : 1, chunk);
    return 2 * align_up((size_t)n_chunks * 2 * d * dtype_size(dtype), 256) + align_up((size_t)(n_chunks + 64) * sizeof(int), 256);
}

// Per-segment (max, sum of exp(x - max)) of data[perm[p], :] over the CSR-ordered positions p, in one pass.
// stats: [n_seg, 2, d] (maxima then sums of each segment side by side).
int32_t narrow_softmax_stats(const void* data, int32_t dtype, int64_t num_rows, int64_t d, const int32_t* row32,
                             const int32_t* perm32, const int64_t* seg_ptr, int64_t n_seg, void* stats,
                             int chunk, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!narrow_softmax_covers(d, dtype)) return fail(PGLAMD_E_SHAPE, "softmax statistics: shape not covered");
    if (ws_bytes < narrow_softmax_workspace_bytes(num_rows, d, dtype, chunk)) return fail(PGLAMD_E_WORKSPACE, "softmax statistics: workspace too small");
    AggParams p{};
    p.x = data; p.out = stats; p.row = row32; p.col = perm32; p.indptr = seg_ptr;
    p.x2 = data; p.x_split = INT32_MAX; p.zero_indptr = seg_ptr;
    p.ldx = d; p.ldo = 2 * d; p.out_rows = n_seg; p.n_csr_rows = n_seg; p.E = (int)num_rows;
    p.chunk = chunk; p.n_chunks = (int)ceil_div(num_rows, chunk); p.align = 1;
    p.tile_cols = (int)d; p.j_base = 0;
    const size_t es = dtype_size(dtype);
    const size_t lv = std::min<size_t>(16, (size_t)d * es);
    p.narrow_vec = (lv & (lv - 1)) == 0 && reinterpret_cast<uintptr_t>(data) % lv == 0;
    const size_t half = align_up((size_t)p.n_chunks * 2 * d * es, 256);
    p.part_head = ws;
    p.part_tail = static_cast<char*>(ws) + half;
    p.long_count = reinterpret_cast<int*>(static_cast<char*>(ws) + 2 * half);
    p.long_list = p.long_count + 64;
    return dtype == PGLAMD_F32 ? softmax_stats_typed<float>(p, st) : softmax_stats_typed<double>(p, st);
}

int32_t launch_narrow(const AggParams& p, int32_t dtype, int rcls, int64_t dy, hipStream_t st, bool* handled) {
    *handled = false;
    int ymode = 0;
    if (p.y) {
        if (dy == 1) ymode = 1;
        else if (dy == p.tile_cols) ymode = 2;
        else return PGLAMD_OK;
    }
    switch (dtype) {
        case PGLAMD_F32: return pick_width<float>(p, dtype, rcls, ymode, st, handled);
        case PGLAMD_F64: return pick_width<double>(p, dtype, rcls, ymode, st, handled);
        case PGLAMD_I32: return pick_width<int32_t>(p, dtype, rcls, ymode, st, handled);
        case PGLAMD_I64: return pick_width<int64_t>(p, dtype, rcls, ymode, st, handled);
        case PGLAMD_F16: return pick_width<__half>(p, dtype, rcls, ymode, st, handled);
        case PGLAMD_BF16: return pick_width<__hip_bfloat16>(p, dtype, rcls, ymode, st, handled);
        default: return PGLAMD_OK;
    }
}

}  // namespace pglamd
