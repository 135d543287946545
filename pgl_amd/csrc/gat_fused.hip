// gat_fused.hip -- K3+K4+K2 in ONE pass over the edges: the whole attention aggregation of GATConv
// (pgl/nn/conv.py:331-339):
//     alpha = send_uv(a_src, a_dst, "add") ; leaky_relu ; edge_softmax (by dst) ; send_ue_recv(f, alpha, "mul", "sum")
// i.e.  out[v, h, :] = sum_{e=(u->v)} softmax_v( leaky(a_src[u,h] + a_dst[v,h]) ) * f[u, h, :]
//
// The reference materialises four [E,H] tensors and makes ~14 passes over them.  Here
// the logits never exist in memory: each wave walks its fixed-size chunk of the dst-sorted edge
// stream (same geometry as agg_flat_kernel) carrying an ONLINE softmax state per lane
// (running max m, running sum s, running weighted row acc; flash-attention style rescaling), so
// HBM traffic is one gather of f[u] (H*D*4 B) + a_src[u] (H*4 B) + 8 B of index per edge and one
// write of the output row: the byte count of a plain SpMM + 6 %.
// Rows that straddle chunk boundaries leave (acc, m, s) partials that a second kernel merges in a
// fixed order with the associative softmax merge  (m, s, a) + (m', s', a') =
// (M = max(m, m'), s e^{m-M} + s' e^{m'-M}, a e^{m-M} + a' e^{m'-M}):  atomic-free, bit-reproducible.
#include "common.hpp"

namespace pglamd {

struct GatParams {
    const float* x; const float* a_src; const float* a_dst; float* out;
    float* row_max; float* row_sum;              // optional [out_rows, H] softmax statistics (NULL to skip)
    const int* row; const int* col; const int64_t* indptr;
    float* part_head; float* part_tail;          // [n_chunks, 3, d]: acc | m | s (m, s replicated per column)
    int* long_count; int* long_list; int* long_list2;   // [2] counters + two-level fix-up work lists
    int64_t out_rows, n_csr_rows;
    int E, n_chunks, chunk, n_blocks, n_grid_chunks;
    int d, H, D;
    float slope;
};

template <int VEC> struct alignas(4 * VEC) FV { float v[VEC]; };

template <int VEC>
__global__ __launch_bounds__(kBlock) void gat_flat_kernel(GatParams p) {
    constexpr int U = 8;
    using V = FV<VEC>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = wave_uniform(threadIdx.x >> 6);
    const int j0 = lane * VEC;
    const bool act = j0 < p.d;
    const int head = act ? j0 / p.D : 0;

    if ((int)blockIdx.x >= p.n_grid_chunks) {           // zero-fill role: rows with no in-edge
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
            if (p.row_max && lane < p.H) { p.row_max[(r0 + l) * p.H + lane] = 0.f; p.row_sum[(r0 + l) * p.H + lane] = 0.f; }
        }
        return;
    }
    const int64_t lb = xcd_swizzle(blockIdx.x, p.n_blocks);
    if (lb < 0) return;
    const int c = wave_uniform((int)lb * kWavesPerBlock + wib);
    if (c >= p.n_chunks) return;
    const cptr<int> rowp = as_const(p.row);
    const cptr<int> colp = as_const(p.col);
    const int e0 = chunk_cut(rowp, as_const(p.indptr), c * p.chunk, p.chunk, p.E);
    const int e1 = chunk_cut(rowp, as_const(p.indptr), c * p.chunk + p.chunk, p.chunk, p.E);
    if (e0 >= e1) return;
    const float* __restrict__ x = p.x;
    const float* __restrict__ asrc = p.a_src;
    const float slope = p.slope;

    float m = -INFINITY, s = 0.f, acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    int cur = rowp[e0];
    bool head_open = e0 > 0 && rowp[e0 - 1] == cur;
    const float* __restrict__ adst = p.a_dst;

    auto store_partial = [&](float* base) {
        if (!act) return;
        float* dst = base + (int64_t)c * 3 * p.d;
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = acc[k];
        *reinterpret_cast<V*>(dst + j0) = o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = m;
        *reinterpret_cast<V*>(dst + p.d + j0) = o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = s;
        *reinterpret_cast<V*>(dst + 2 * p.d + j0) = o;
    };
    auto store_final = [&](int r) {
        if (r >= p.out_rows || !act) return;
        V o;
        const float inv = 1.f / s;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = acc[k] * inv;
        *reinterpret_cast<V*>(p.out + (int64_t)r * p.d + j0) = o;
        if (p.row_max && (j0 % p.D) == 0) { p.row_max[(int64_t)r * p.H + head] = m; p.row_sum[(int64_t)r * p.H + head] = s; }
    };
    // a_dst[row] rides along with every edge of the batch (an L1/L2 hit after the first edge of a
    // row) instead of being fetched on the row change, which would stall the wave once per row.
    auto consume = [&](int r, float as_val, float ad, const V& xv) {
        if (r != cur) {
            if (head_open) store_partial(p.part_head); else store_final(cur);
            head_open = false;
            cur = r;
            m = -INFINITY; s = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        }
        float l = as_val + ad;
        l = l > 0.f ? l : slope * l;
        const float mn = fmaxf(m, l);
        const float sc = expf(m - mn);          // m = -inf on the first edge of a row: e^{-inf} = 0
        const float pe = expf(l - mn);
        s = s * sc + pe;
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] * sc + pe * xv.v[k];
        m = mn;
    };
    auto load_idx = [&](int e, int (&cc)[U], int (&rr)[U]) {
#pragma unroll
        for (int i = 0; i < U; ++i) { rr[i] = rowp[e + i]; cc[i] = colp[e + i]; }
    };
    auto load_rows = [&](const int (&cc)[U], const int (&rr)[U], V (&vx)[U], float (&va)[U], float (&vd)[U]) {
#pragma unroll
        for (int i = 0; i < U; ++i)
            if (act) {
                vx[i] = *reinterpret_cast<const V*>(x + (int64_t)cc[i] * p.d + j0);
                va[i] = asrc[(int64_t)cc[i] * p.H + head];
                vd[i] = adst[(int64_t)rr[i] * p.H + head];
            }
    };

    int e = e0;
    const int n_full = (e1 - e0) / U;
    int cA[U], rA[U]; V xA[U]; float aA[U], dA[U];
    if (n_full > 0) { load_idx(e, cA, rA); load_rows(cA, rA, xA, aA, dA); }
    for (int g = 0; g < n_full; ++g) {
        int cB[U], rB[U]; V xB[U]; float aB[U], dB[U];
        const bool more = g + 1 < n_full;
        if (more) { load_idx(e + U, cB, rB); load_rows(cB, rB, xB, aB, dB); }
#pragma unroll
        for (int i = 0; i < U; ++i) consume(rA[i], aA[i], dA[i], xA[i]);
        if (more) {
#pragma unroll
            for (int i = 0; i < U; ++i) { rA[i] = rB[i]; xA[i] = xB[i]; aA[i] = aB[i]; dA[i] = dB[i]; }
        }
        e += U;
    }
    for (; e < e1; ++e) {
        const int r = rowp[e], cc = colp[e];
        V xv{}; float av = 0.f, dv = 0.f;
        if (act) {
            xv = *reinterpret_cast<const V*>(x + (int64_t)cc * p.d + j0);
            av = asrc[(int64_t)cc * p.H + head];
            dv = adst[(int64_t)r * p.H + head];
        }
        consume(r, av, dv, xv);
    }
    const bool tail_open = e1 < p.E && rowp[e1] == cur;
    if (head_open) store_partial(p.part_head);
    else if (tail_open) {
        store_partial(p.part_tail);
        if (lane == 0) p.long_list[atomicAdd(p.long_count, 1)] = c;
    } else store_final(cur);
}

// merges the (acc, m, s) partials of the rows longer than a chunk (only those are split), from the
// work list the flat kernel filled.  Pass 1 (LONG = false): one wave per task, rows with <= 16
// partials are merged right there, longer ones go to a second list.  Pass 2 (LONG = true): 1024-thread
// blocks, 16 waves split one hub row's partial list, LDS combine in wave order.  The softmax merge is
// associative; every row's own merge order is fixed => bit-reproducible.
constexpr int kGatFixShort = 16;
constexpr int kGatFixWaves = 16;
constexpr int kGatFixGridShort = 2048;
constexpr int kGatFixGridLong = 512;

template <int VEC, bool LONG>
__global__ __launch_bounds__(LONG ? kGatFixWaves * kWave : kBlock) void gat_fixup_kernel(GatParams p) {
    using V = FV<VEC>;
    constexpr int NW = LONG ? kGatFixWaves : 1;
    __shared__ float red[LONG ? kGatFixWaves : 1][3][LONG ? kWave * VEC : 1];
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
        float m = -INFINITY, s = 0.f, acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        auto merge_vals = [&](const V& va, float m2, float s2) {
            const float mn = fmaxf(m, m2);
            const float c1 = expf(m - mn), c2 = expf(m2 - mn);
            s = s * c1 + s2 * c2;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] * c1 + va.v[k] * c2;
            m = mn;
        };
        auto merge = [&](const float* base) {
            merge_vals(*reinterpret_cast<const V*>(base + j0), base[p.d + j0], base[2 * p.d + j0]);
        };
        if constexpr (!LONG) {
            if (act) {
                merge(p.part_tail + (int64_t)a * 3 * p.d);
#pragma unroll 4
                for (int c = a + 1; c <= b; ++c) merge(p.part_head + (int64_t)c * 3 * p.d);
            }
        } else {
            if (act)
                for (int c = a + 1 + wib; c <= b; c += NW) merge(p.part_head + (int64_t)c * 3 * p.d);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VEC; ++k) red[wib][0][lane * VEC + k] = acc[k];
            red[wib][1][lane * VEC] = m; red[wib][2][lane * VEC] = s;
            __syncthreads();
            if (wib != 0) continue;
            // wave 0: tail partial of chunk a first, then the wave results in wave order
            m = -INFINITY; s = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
            if (act) {
                merge(p.part_tail + (int64_t)a * 3 * p.d);
                for (int w = 0; w < NW; ++w) {
                    const float s2 = red[w][2][lane * VEC];
                    if (s2 == 0.f) continue;                     // that wave had no partial
                    V va;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) va.v[k] = red[w][0][lane * VEC + k];
                    merge_vals(va, red[w][1][lane * VEC], s2);
                }
            }
        }
        if (!act || r >= p.out_rows) continue;
        V o;
        const float inv = 1.f / s;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = acc[k] * inv;
        *reinterpret_cast<V*>(p.out + (int64_t)r * p.d + j0) = o;
        if (p.row_max && (j0 % p.D) == 0) { p.row_max[(int64_t)r * p.H + j0 / p.D] = m; p.row_sum[(int64_t)r * p.H + j0 / p.D] = s; }
    }
}

static int gat_chunk_edges() {
    static int k = [] {
        const char* s = getenv("PGLAMD_CHUNK");
        int v = s ? atoi(s) : 256;
        if (v < 8) v = 8;
        return v / 8 * 8;
    }();
    return k;
}

template <int VEC>
static int32_t launch_gat(GatParams p, hipStream_t st) {
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    const int64_t zb = ceil_div(ceil_div(p.out_rows, kWave), kWavesPerBlock);
    if (p.n_chunks > 1) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(gat_flat_kernel<VEC>, dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
    PGLAMD_LAUNCH_CHECK();
    if (p.n_chunks > 1) {
        hipLaunchKernelGGL((gat_fixup_kernel<VEC, false>), dim3(kGatFixGridShort), dim3(kBlock), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
        hipLaunchKernelGGL((gat_fixup_kernel<VEC, true>), dim3(kGatFixGridLong), dim3(kGatFixWaves * kWave), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
    }
    return PGLAMD_OK;
}

}  // namespace pglamd

using namespace pglamd;

extern "C" size_t pglamd_gat_aggregate_workspace_bytes(int64_t num_edges, int64_t heads, int64_t head_dim) {
    if (num_edges <= 0) return 256;
    const int64_t n_chunks = ceil_div(num_edges, gat_chunk_edges());
    return 2 * align_up((size_t)n_chunks * 3 * heads * head_dim * sizeof(float), 256) +
           2 * align_up((size_t)(n_chunks + 64) * sizeof(int), 256) + 256;
}

extern "C" int32_t pglamd_gat_aggregate(const float* feature, const float* attn_src, const float* attn_dst, int64_t heads,
                                        int64_t head_dim, float negative_slope, const int32_t* row, const int32_t* col,
                                        const int64_t* indptr, int64_t num_edges, int64_t n_csr_rows, int64_t out_rows,
                                        float* out, float* row_max, float* row_sum, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    if (!out || !indptr || heads <= 0 || head_dim <= 0 || out_rows < 0 || (num_edges > 0 && (!feature || !attn_src || !attn_dst || !row || !col)))
        return fail(PGLAMD_E_ARG, "gat_aggregate: bad argument");
    if ((row_max == nullptr) != (row_sum == nullptr)) return fail(PGLAMD_E_ARG, "gat_aggregate: row_max/row_sum must both be given or both NULL");
    if (num_edges < 0 || num_edges >= INT32_MAX || out_rows >= INT32_MAX) return fail(PGLAMD_E_RANGE, "gat_aggregate: sizes beyond int32 engine range");
    const int64_t d = heads * head_dim;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out_rows == 0) return PGLAMD_OK;
    if (num_edges == 0) {
        PGLAMD_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)out_rows * d * sizeof(float), st));
        if (row_max) {
            PGLAMD_HIP_CHECK(hipMemsetAsync(row_max, 0, (size_t)out_rows * heads * sizeof(float), st));
            PGLAMD_HIP_CHECK(hipMemsetAsync(row_sum, 0, (size_t)out_rows * heads * sizeof(float), st));
        }
        return PGLAMD_OK;
    }
    // lane geometry: one 64-lane tile must cover all H*D columns, VEC elements of ONE head per lane
    int vec = 0;
    const uintptr_t al = reinterpret_cast<uintptr_t>(feature) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(workspace);
    for (int v = 1; v <= 4; v <<= 1)
        if (d % v == 0 && head_dim % v == 0 && d / v <= kWave && al % (4 * v) == 0) { vec = v; break; }
    if (vec == 0 || heads > kWave)
        return fail(PGLAMD_E_SHAPE, "gat_aggregate: heads*head_dim = %lld does not fit one 64-lane tile (max 256 with head_dim %% 4 == 0)", (long long)d);
    if (!workspace || workspace_bytes < pglamd_gat_aggregate_workspace_bytes(num_edges, heads, head_dim))
        return fail(PGLAMD_E_WORKSPACE, "gat_aggregate: workspace too small");
    GatParams p{};
    p.x = feature; p.a_src = attn_src; p.a_dst = attn_dst; p.out = out; p.row_max = row_max; p.row_sum = row_sum;
    p.row = row; p.col = col; p.indptr = indptr;
    p.out_rows = out_rows; p.n_csr_rows = n_csr_rows; p.E = (int)num_edges;
    p.chunk = gat_chunk_edges(); p.n_chunks = (int)ceil_div(num_edges, p.chunk);
    p.d = (int)d; p.H = (int)heads; p.D = (int)head_dim; p.slope = negative_slope;
    const size_t half = align_up((size_t)p.n_chunks * 3 * d * sizeof(float), 256);
    p.part_head = static_cast<float*>(workspace);
    p.part_tail = reinterpret_cast<float*>(static_cast<char*>(workspace) + half);
    p.long_count = reinterpret_cast<int*>(static_cast<char*>(workspace) + 2 * half);
    p.long_list = p.long_count + 64;
    p.long_list2 = reinterpret_cast<int*>(static_cast<char*>(workspace) + 2 * half + align_up((size_t)(p.n_chunks + 64) * sizeof(int), 256));
    switch (vec) {
        case 1: return launch_gat<1>(p, st);
        case 2: return launch_gat<2>(p, st);
        default: return launch_gat<4>(p, st);
    }
}
