// grad_ops.hip -- gradient kernels for the two paths pgl_amd/autograd.py used to compose from [E, d] row gathers (round 3):
//
//   pglamd_winner_grad        d x of send_recv(x, max | min): the gradient of an output row flows to every message equal to
//                             the winner (Paddle's rule, graph_send_recv_grad):
//                                 gx[u, j] = sum_{e = (u -> v)} [x[u, j] == out[v, j]] * g[v, j]
//                             ONE walk of the src-sorted stream; x[u] is the row operand (registers), g[v] and out[v] are gathered
//                             per edge; no [E, d] tensor (the composition materialised five of them).
//   pglamd_edge_operand_grad  d y of send_ue_recv(x, y, mop, sum | mean) for trailing-dim broadcast operands other than the
//                             [E, H, 1] case the SDDMM kernel covers (y [E], [E, 1], [E, d], [E, H, D]):
//                                 gy[e, jy] = sum_{j in group jy} s[v] g[v, j] * { x[u, j] (mul) | 1 (add) | -1 (sub) | -x[u, j] / y^2 (div) }
//                             one wave per edge slot, lanes across the columns, group sums by DPP-free xor shuffles.
//
// Both replace reference behaviour that PaddlePaddle implements inside graph_send_recv_grad / graph_send_ue_recv_grad
// (pgl/graph.py:834-937 call sites); fp32 only -- other dtypes keep the composed path.
#include "aggregate_flat.hpp"

namespace pglamd {

// ---- winner gradient: chunked walk (same chunk_cut / partial protocol as the flat kernel, simple 4-edge unroll) ---------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void winner_grad_kernel(AggParams p, const float* __restrict__ winner) {
    constexpr int U = 4;
    using V = VecT<float, VEC>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    if ((int)blockIdx.x >= p.n_grid_chunks) { zero_empty_rows_role<float>(p, (int64_t)blockIdx.x - p.n_grid_chunks, lane); return; }
    const int64_t lb = xcd_swizzle(blockIdx.x, p.n_blocks);
    if (lb < 0) return;
    const int c = wave_uniform((int)lb * kWavesPerBlock + wib);
    if (c >= p.n_chunks) return;
    const cptr<int> rowp = as_const(p.row);
    const cptr<int> colp = as_const(p.col);
    const int e0 = chunk_cut(rowp, as_const(p.indptr), c * p.chunk, p.chunk, p.E);
    const int e1 = chunk_cut(rowp, as_const(p.indptr), c * p.chunk + p.chunk, p.chunk, p.E);
    if (e0 >= e1) return;
    const float* __restrict__ g = static_cast<const float*>(p.x);          // gathered by column id: upstream gradient rows
    const float* __restrict__ xr = static_cast<const float*>(p.y);         // row operand: the forward's input rows
    const int j0 = lane * VEC;
    const bool act = j0 < p.tile_cols;
    float acc[VEC];
    V xrow{};
    int cur = rowp[e0];
    bool head_open = e0 > 0 && rowp[e0 - 1] == cur;
    auto open_row = [&](int r) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        if (act) xrow = *reinterpret_cast<const V*>(xr + (int64_t)r * p.ldy + j0);
    };
    auto store_partial = [&](bool head) {
        float* dst = static_cast<float*>(head ? p.part_head : p.part_tail) + (int64_t)c * p.tile_cols;
        if (act) { V o; for (int k = 0; k < VEC; ++k) o.v[k] = acc[k]; *reinterpret_cast<V*>(dst + j0) = o; }
        if (!head && lane == 0) p.long_list[atomicAdd(p.long_count, 1)] = c;
    };
    auto store_final = [&](int r) {
        if (r >= p.out_rows || !act) return;
        V o; for (int k = 0; k < VEC; ++k) o.v[k] = acc[k];
        *reinterpret_cast<V*>(static_cast<float*>(p.out) + (int64_t)r * p.ldo + j0) = o;
    };
    auto consume = [&](int r, const V& gv, const V& wv) {
        if (r != cur) {
            if (head_open) store_partial(true); else store_final(cur);
            head_open = false; cur = r; open_row(r);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += (wv.v[k] == xrow.v[k]) ? gv.v[k] : 0.f;
    };
    open_row(cur);
    int e = e0;
    for (; e + U <= e1; e += U) {
        int rr[U], cc[U];
        V gv[U], wv[U];
#pragma unroll
        for (int i = 0; i < U; ++i) { rr[i] = rowp[e + i]; cc[i] = colp[e + i]; }
#pragma unroll
        for (int i = 0; i < U; ++i)
            if (act) {
                gv[i] = *reinterpret_cast<const V*>(g + (int64_t)cc[i] * p.ldx + j0);
                wv[i] = *reinterpret_cast<const V*>(winner + (int64_t)cc[i] * p.ldx + j0);
            }
#pragma unroll
        for (int i = 0; i < U; ++i) consume(rr[i], gv[i], wv[i]);
    }
    for (; e < e1; ++e) {
        const int r = rowp[e], cc = colp[e];
        V gv{}, wv{};
        if (act) { gv = *reinterpret_cast<const V*>(g + (int64_t)cc * p.ldx + j0); wv = *reinterpret_cast<const V*>(winner + (int64_t)cc * p.ldx + j0); }
        consume(r, gv, wv);
    }
    const bool tail_open = e1 < p.E && rowp[e1] == cur;
    if (head_open) store_partial(true);
    else if (tail_open) store_partial(false);
    else store_final(cur);
}

template <int VEC>
static int32_t launch_winner(AggParams p, const float* winner, hipStream_t st) {
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    const int64_t zb = ceil_div(ceil_div(p.out_rows, kWave), kWavesPerBlock);
    if (p.n_chunks > 1) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL((winner_grad_kernel<VEC>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p, winner);
    PGLAMD_LAUNCH_CHECK();
    if (p.n_chunks > 1) {
        hipLaunchKernelGGL((agg_fixup_kernel<float, VEC, 1, 0, false>), dim3((unsigned)std::min<int64_t>(kFixGridShort, ceil_div(p.n_chunks, kWavesPerBlock))), dim3(kBlock), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
        hipLaunchKernelGGL((agg_fixup_kernel<float, VEC, 1, 0, true>), dim3((unsigned)std::min<int64_t>(kFixGridLong, p.n_chunks)), dim3(kFixWaves * kWave), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
    }
    return PGLAMD_OK;
}

// ---- edge-operand gradient: one wave per edge slot --------------------------------------------------------------------------
// lanes across the d columns (VEC per lane); a group of `glanes` consecutive lanes shares one y element
template <int VEC>
__global__ __launch_bounds__(kBlock) void edge_operand_grad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                   const float* __restrict__ y, const float* __restrict__ dscale,
                                                                   const int* __restrict__ row, const int* __restrict__ col,
                                                                   const int* __restrict__ eid, int64_t E, int d, int dy, int glanes,
                                                                   int mop, float* __restrict__ out) {
    constexpr int U = 4;
    using V = VecT<float, VEC>;
    const int lane = threadIdx.x & (kWave - 1);
    const int j0 = lane * VEC;
    const bool act = j0 < d;
    const int gcols = d / dy;                       // columns per y element
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t p0 = wave * U; p0 < E; p0 += n_waves * U) {
        V gv[U], xv[U];
        int oe[U];
        float ds[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int64_t p = p0 + i;
            oe[i] = -1;
            if (p < E) {
                const int r = row[p], c = col[p];
                oe[i] = eid ? eid[p] : (int)p;
                ds[i] = dscale ? dscale[r] : 1.f;
                if (act) {
                    gv[i] = *reinterpret_cast<const V*>(g + (int64_t)r * d + j0);
                    if (mop >= PGLAMD_MUL) xv[i] = *reinterpret_cast<const V*>(x + (int64_t)c * d + j0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (oe[i] < 0) continue;                 // wave-uniform
            float part[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float v = act ? gv[i].v[k] * ds[i] : 0.f;
                if (mop == PGLAMD_MUL) v *= act ? xv[i].v[k] : 0.f;
                else if (mop == PGLAMD_DIV) v *= act ? -xv[i].v[k] : 0.f;
                else if (mop == PGLAMD_SUB) v = -v;
                part[k] = v;
            }
            float* orow = out + (int64_t)oe[i] * dy;
            if (gcols >= VEC) {
                // the lane's VEC columns lie in ONE group: sum them, then across the group's lanes
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < VEC; ++k) s += part[k];
                for (int off = 1; off < glanes; off <<= 1) s += __shfl_xor(s, off, kWave);
                if (act && (lane & (glanes - 1)) == 0) {
                    const int jy = j0 / gcols;
                    if (mop == PGLAMD_DIV) { const float yy = y[(int64_t)oe[i] * dy + jy]; s = s / (yy * yy); }
                    orow[jy] = s;
                }
            } else {
                // several y elements per lane (gcols < VEC, incl. elementwise gcols == 1)
                if (act) {
#pragma unroll
                    for (int k0 = 0; k0 < VEC; k0 += 1) {
                        if ((k0 % gcols) != 0) continue;
                        float s = 0.f;
                        for (int k = k0; k < k0 + gcols && k < VEC; ++k) s += part[k];
                        const int jy = (j0 + k0) / gcols;
                        if (mop == PGLAMD_DIV) { const float yy = y[(int64_t)oe[i] * dy + jy]; s = s / (yy * yy); }
                        orow[jy] = s;
                    }
                }
            }
        }
    }
}

}  // namespace pglamd

using namespace pglamd;

extern "C" size_t pglamd_winner_grad_workspace_bytes(int64_t num_edges, int64_t d) {
    return pglamd_aggregate_workspace_bytes(num_edges, d, PGLAMD_F32);
}

extern "C" int32_t pglamd_winner_grad(const float* grad_out, const float* out, const float* x, int64_t d, const int32_t* src_row,
                                      const int32_t* src_col, const int64_t* src_indptr, int64_t num_edges, int64_t n_x_rows,
                                      float* grad_x, void* workspace, size_t workspace_bytes, void* stream) {
    if (!grad_x || !src_indptr || (num_edges > 0 && (!grad_out || !out || !x || !src_row || !src_col)))
        return fail(PGLAMD_E_ARG, "winner_grad: NULL pointer");
    if (d <= 0 || d > 256) return fail(PGLAMD_E_SHAPE, "winner_grad: rows of 1..256 fp32 columns (got %lld)", (long long)d);
    if (num_edges < 0 || num_edges > kMaxEdges || n_x_rows >= INT32_MAX) return fail(PGLAMD_E_RANGE, "winner_grad: sizes beyond int32 engine range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (num_edges == 0) return zero_empty_rows(src_indptr, n_x_rows, n_x_rows, grad_x, (size_t)d * 4, st);
    const uintptr_t al = reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x) |
                         reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(workspace);
    int vec = d > 128 ? 4 : d > 64 ? 2 : 1;
    while (vec > 1 && (d % vec != 0 || al % (vec * 4) != 0)) vec >>= 1;
    if ((int64_t)vec * kWave < d) return fail(PGLAMD_E_SHAPE, "winner_grad: d=%lld needs %d-element lanes but is not aligned for them", (long long)d, d > 128 ? 4 : 2);
    AggParams p{};
    p.x = grad_out; p.x2 = grad_out; p.x_split = INT32_MAX; p.y = x; p.out = grad_x; p.row = src_row; p.col = src_col;
    p.indptr = src_indptr; p.zero_indptr = src_indptr;
    p.ldx = d; p.ldy = d; p.ldo = d; p.out_rows = n_x_rows; p.n_csr_rows = n_x_rows; p.E = (int)num_edges;
    p.j_base = 0; p.tile_cols = (int)d; p.zvec = vec; p.align = 1; p.gy = 1;
    p.chunk = chunk_edges_for(num_edges);
    p.n_chunks = (int)ceil_div(num_edges, (int64_t)p.chunk);
    const size_t half = align_up((size_t)p.n_chunks * d * sizeof(float), 256);
    const size_t lst = align_up((size_t)(p.n_chunks + 64) * sizeof(int), 256);
    if (!workspace || workspace_bytes < 2 * half + 2 * lst) return fail(PGLAMD_E_WORKSPACE, "winner_grad: workspace too small");
    char* ws = static_cast<char*>(workspace);
    p.part_head = ws; p.part_tail = ws + half;
    p.long_count = reinterpret_cast<int*>(ws + 2 * half);
    p.long_list = p.long_count + 64;
    p.long_list2 = reinterpret_cast<int*>(ws + 2 * half + lst);
    return vec == 4 ? launch_winner<4>(p, out, st) : vec == 2 ? launch_winner<2>(p, out, st) : launch_winner<1>(p, out, st);
}

extern "C" int32_t pglamd_edge_operand_grad(const float* grad_out, const float* x, const float* y, const float* dst_scale,
                                            int64_t d, int64_t dy, const int32_t* row, const int32_t* col, const int32_t* eid,
                                            int64_t num_edges, int32_t message_op, float* grad_y, void* stream) {
    if (num_edges < 0 || d <= 0 || dy <= 0 || d % dy != 0 || d > 256) return fail(PGLAMD_E_SHAPE, "edge_operand_grad: d=%lld dy=%lld", (long long)d, (long long)dy);
    if (message_op < PGLAMD_ADD || message_op > PGLAMD_DIV) return fail(PGLAMD_E_ARG, "edge_operand_grad: bad op enum");
    if (num_edges == 0) return PGLAMD_OK;
    if (!grad_out || !grad_y || !row || !col || (message_op >= PGLAMD_MUL && !x) || (message_op == PGLAMD_DIV && !y))
        return fail(PGLAMD_E_ARG, "edge_operand_grad: NULL pointer");
    const uintptr_t al = reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(x);
    int vec = d > 128 ? 4 : d > 64 ? 2 : 1;
    while (vec > 1 && (d % vec != 0 || al % (vec * 4) != 0)) vec >>= 1;
    if ((int64_t)vec * kWave < d) return fail(PGLAMD_E_SHAPE, "edge_operand_grad: d=%lld is not aligned for wide lanes", (long long)d);
    const int gcols = (int)(d / dy);
    int glanes = 1;
    if (gcols >= vec) {
        if (gcols % vec != 0) return fail(PGLAMD_E_SHAPE, "edge_operand_grad: group of %d columns vs %d-element lanes", gcols, vec);
        glanes = gcols / vec;
        if (glanes > kWave) glanes = kWave;
        if ((glanes & (glanes - 1)) != 0 || (int64_t)glanes * vec != gcols)
            return fail(PGLAMD_E_SHAPE, "edge_operand_grad: a group must span a power-of-two number of lanes (group %d columns)", gcols);
    } else if (vec % gcols != 0) {
        return fail(PGLAMD_E_SHAPE, "edge_operand_grad: group of %d columns vs %d-element lanes", gcols, vec);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(ceil_div(num_edges, (int64_t)4), kWavesPerBlock), 256 * 64);
#define LAUNCH(V) hipLaunchKernelGGL((edge_operand_grad_kernel<V>), dim3(grid), dim3(kBlock), 0, st, grad_out, x, y, dst_scale, row, col, eid, \
                                     num_edges, (int)d, (int)dy, glanes, message_op, grad_y)
    if (vec == 4) LAUNCH(4); else if (vec == 2) LAUNCH(2); else LAUNCH(1);
#undef LAUNCH
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}
