// aggregate.hip -- K1/K2: atomic-free segmented aggregation over a CSR-ordered edge list.
//
//   out[r, :] = dst_scale[r] * REDUCE_{p: row[p]==r} ( src_scale[col[p]] * x[col[p], :] (mop) y[yp, :] )
//
// Replaces paddle.geometric.send_u_recv / send_ue_recv (pgl/graph.py:859-861, 885-887, 929-937)
// and, with col == NULL (identity gather), paddle.geometric.segment_* (pgl/math.py:30-178).
//
// Design (MI355X, HBM-bound: ~516 B of gathered feature row per edge at d=128 fp32, 2 flops/B^-1):
//   * The edge list is a flat stream sorted by destination row.  Wave w owns the fixed-size chunk
//     [w*K, (w+1)*K) of that stream regardless of row boundaries (merge-path style): perfect
//     load balance on power-law graphs, no per-row launch geometry, no atomics.
//   * Inside a chunk everything that steers control flow is WAVE-UNIFORM: col/row/eid indices are
//     read with scalar loads (s_load_dwordx8), the feature row address is an SGPR base + lane
//     offset, row-boundary tests are scalar branches.  The 64 lanes span the feature dimension
//     (VEC contiguous elements per lane, NT tiles per lane): one gathered row = one fully
//     coalesced wave-wide load (512 B for d=128 fp32).
//   * U = 8 edges are issued back-to-back and double-buffered, so a wave keeps 8..16 independent
//     row gathers (4-8 KiB) in flight irrespective of how short the rows are; at 8 waves/SIMD
//     that is >128 KiB per CU outstanding, enough to cover HBM latency at >5 TB/s.
//   * A row that lies wholly inside a chunk is reduced in registers and stored once.  A row that
//     straddles chunk boundaries leaves per-chunk partials in the caller's workspace
//     (tail partial T[c] for the chunk where it starts, head partial H[c] for every later chunk)
//     and a second tiny kernel adds them IN CHUNK ORDER: deterministic, bit-reproducible run
//     to run, every output row written exactly once.  Rows without edges are zero-filled by a
//     third kernel from indptr (the reference guarantees 0, not +-inf, for every reduce op).
//   * Logical blocks are remapped so consecutive chunks (consecutive destination rows) run on
//     the same XCD: partition-ordered graphs then reuse source rows in that XCD's private L2.
#include "aggregate_flat.hpp"
#include "aggregate_dense2.hpp"

namespace pglamd {

int chunk_edges() {
    static int k = [] {
        const char* s = getenv("PGLAMD_CHUNK");
        int v = s ? atoi(s) : 256;
        if (v < 8) v = 8;
        return v / 8 * 8;
    }();
    return k;
}

// Edges per chunk for an edge stream of E edges.  256 is the optimum at benchmark size (20 M edges: 78 k chunks, ten waves per
// wave slot of the chip); a stream of a few million edges -- one rank's share of a partitioned graph, a sampled block -- cut
// at 256 gives barely one wave per slot and the launch ends with the slowest wave.  Measured on rank 0 of the 8-way row
// partition of the benchmark graph (2.6 M edges in three launches): 0.384 ms at 256, 0.335 at 128, 0.301 at 64; at 10 M edges
// 0.851 / 0.834.  PGLAMD_CHUNK pins one value (stress tests).
int chunk_edges_for(int64_t num_edges) {
    if (getenv("PGLAMD_CHUNK")) return chunk_edges();
    return num_edges >= 12000000 ? 256 : num_edges >= 5000000 ? 128 : 64;
}

// Optional in-library timing of the dominant kernel (bench.py's roofline leg): while enabled,
// every flat-kernel launch is bracketed by a pair of HIP events on the launch stream.
ProfileState& prof() { static ProfileState s; return s; }

template <typename T>
static int32_t launch_fixup_typed(const AggParams& p, int rcls, hipStream_t st) {
    const unsigned gs = (unsigned)std::min<int64_t>(kFixGridShort, ceil_div(p.n_chunks, kWavesPerBlock));
    const unsigned gl = (unsigned)std::min<int64_t>(kFixGridLong, p.n_chunks);
    if (rcls == 0) {
        hipLaunchKernelGGL((agg_fixup_kernel<T, 1, 1, 0, false>), dim3(gs), dim3(kBlock), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
        hipLaunchKernelGGL((agg_fixup_kernel<T, 1, 1, 0, true>), dim3(gl), dim3(kFixWaves * kWave), 0, st, p);
    } else {
        hipLaunchKernelGGL((agg_fixup_kernel<T, 1, 1, 1, false>), dim3(gs), dim3(kBlock), 0, st, p);
        PGLAMD_LAUNCH_CHECK();
        hipLaunchKernelGGL((agg_fixup_kernel<T, 1, 1, 1, true>), dim3(gl), dim3(kFixWaves * kWave), 0, st, p);
    }
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

int32_t launch_fixup_cols(const AggParams& p, int32_t dtype, int rcls, hipStream_t st) {
    switch (dtype) {
        case PGLAMD_F32: return launch_fixup_typed<float>(p, rcls, st);
        case PGLAMD_F64: return launch_fixup_typed<double>(p, rcls, st);
        case PGLAMD_I32: return launch_fixup_typed<int32_t>(p, rcls, st);
        case PGLAMD_I64: return launch_fixup_typed<int64_t>(p, rcls, st);
        case PGLAMD_F16: return launch_fixup_typed<__half>(p, rcls, st);
        case PGLAMD_BF16: return launch_fixup_typed<__hip_bfloat16>(p, rcls, st);
        default: return fail(PGLAMD_E_DTYPE, "fix-up: dtype %d", dtype);
    }
}

int narrow_max() {
    static int k = [] { const char* s = getenv("PGLAMD_NARROW"); return s ? atoi(s) : 16; }();
    return k;
}
// The lane-per-edge kernels walk 64 edges per step, so they amortise their per-chunk prologue over longer chunks:
// 512 edges measured best at C2 sizes (d=1: 0.21 -> 0.13 ms against 256; 1024 equal, 4096 slower).
int narrow_chunk_edges() {
    static int k = [] {
        const char* s = getenv("PGLAMD_NCHUNK");
        int v = s ? atoi(s) : 512;
        if (getenv("PGLAMD_CHUNK")) v = chunk_edges();       // stress tests drive both kernels with one knob
        return v < 8 ? 8 : v / 8 * 8;
    }();
    return k;
}

// aggregate_group.hpp: a wave walks 1024 edges whatever its group count (128 per 8-lane group ... 512 per 32-lane group);
// measured at C2, d = 32 fp32: 256 -> 0.70 ms, 512 -> 0.44, 1024 -> 0.33, 2048 -> 0.34
int group_wave_edges() {
    static int k = [] { const char* s = getenv("PGLAMD_GCHUNK"); int v = s ? atoi(s) : 1024; return v < 64 ? 64 : v; }();
    return k;
}
int group_row_bytes() {
    static int k = [] { const char* s = getenv("PGLAMD_GROUP_BYTES"); int v = s ? atoi(s) : 256; return v > 256 ? 256 : v; }();
    return k;
}
int group_min_bytes() {
    static int k = [] { const char* s = getenv("PGLAMD_GROUP_MIN_BYTES"); int v = s ? atoi(s) : 64; return v < 16 ? 16 : v; }();
    return k;
}
int group_chunk_edges(int groups_per_wave) {
    if (getenv("PGLAMD_CHUNK")) return chunk_edges();         // stress tests drive every kernel with one knob
    const int v = group_wave_edges() / groups_per_wave;
    return v < 8 ? 8 : v / 8 * 8;
}

int32_t zero_empty_rows(const int64_t* indptr, int64_t n_csr_rows, int64_t out_rows, void* out,
                               size_t row_bytes, hipStream_t st) {
    if (out_rows <= 0 || row_bytes == 0) return PGLAMD_OK;
    const int64_t waves = ceil_div(out_rows, kWave);
    const unsigned grid = (unsigned)ceil_div(waves, kWavesPerBlock);
    const uintptr_t a = reinterpret_cast<uintptr_t>(out);
    if (row_bytes % 16 == 0 && a % 16 == 0)
        hipLaunchKernelGGL(zero_empty_rows_kernel<uint4>, dim3(grid), dim3(kBlock), 0, st, indptr, n_csr_rows, out_rows, static_cast<uint4*>(out), (int64_t)(row_bytes / 16));
    else if (row_bytes % 8 == 0 && a % 8 == 0)
        hipLaunchKernelGGL(zero_empty_rows_kernel<uint2>, dim3(grid), dim3(kBlock), 0, st, indptr, n_csr_rows, out_rows, static_cast<uint2*>(out), (int64_t)(row_bytes / 8));
    else if (row_bytes % 4 == 0 && a % 4 == 0)
        hipLaunchKernelGGL(zero_empty_rows_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, st, indptr, n_csr_rows, out_rows, static_cast<uint32_t*>(out), (int64_t)(row_bytes / 4));
    else
        hipLaunchKernelGGL(zero_empty_rows_kernel<uint16_t>, dim3(grid), dim3(kBlock), 0, st, indptr, n_csr_rows, out_rows, static_cast<uint16_t*>(out), (int64_t)(row_bytes / 2));
    PGLAMD_LAUNCH_CHECK();
    return PGLAMD_OK;
}

// float here; the other storage types are instantiated in aggregate_f64.hip / aggregate_more.hip / aggregate_half.hip
template int32_t aggregate_typed<float>(PGLAMD_AGG_ARGS);
extern template int32_t aggregate_typed<double>(PGLAMD_AGG_ARGS);
extern template int32_t aggregate_typed<int32_t>(PGLAMD_AGG_ARGS);
extern template int32_t aggregate_typed<int64_t>(PGLAMD_AGG_ARGS);
extern template int32_t aggregate_typed<__half>(PGLAMD_AGG_ARGS);
extern template int32_t aggregate_typed<__hip_bfloat16>(PGLAMD_AGG_ARGS);

}  // namespace pglamd

using namespace pglamd;

extern "C" size_t pglamd_aggregate_workspace_bytes(int64_t num_edges, int64_t dout, int32_t dtype) {
    size_t es = dtype_size(dtype);
    if (es == 0 || num_edges <= 0) return 256;
    int64_t k = chunk_edges_for(num_edges);
    if ((int64_t)(dout * (int64_t)es) <= group_row_bytes()) k = std::min<int64_t>(k, group_chunk_edges(16));   // grouped kernel: shorter chunks
    const int64_t n_chunks = ceil_div(num_edges, k);
    const int64_t max_cols = 1024;                      // widest tile any (VEC, NT) pair covers
    const int64_t tile = dout < max_cols ? dout : max_cols;
    if (es == 2) es = 4;                                // 16-bit floats keep fp32 partials
    return 2 * align_up((size_t)n_chunks * tile * es, 256) + 2 * align_up((size_t)(n_chunks + 64) * sizeof(int), 256) + 256;
}

static int32_t aggregate_entry(const void* x, int32_t dtype, int64_t dx, const void* y, int64_t dy, const int32_t* eid,
                               const int32_t* row, const int32_t* col, const int64_t* indptr, int64_t num_edges,
                               int64_t n_csr_rows, int64_t out_rows, int64_t dout, int32_t message_op, int32_t reduce_op,
                               const float* src_scale, const float* dst_scale, int32_t accumulate, void* out, void* workspace,
                               size_t workspace_bytes, const AggExtra& ex, void* stream) {
    if (!out || !indptr || (num_edges > 0 && (!x || !row))) return fail(PGLAMD_E_ARG, "aggregate: NULL pointer");
    if (num_edges < 0 || num_edges > kMaxEdges || n_csr_rows >= INT32_MAX || out_rows >= INT32_MAX)
        return fail(PGLAMD_E_RANGE, "aggregate: sizes beyond int32 engine range");
    if (dout <= 0 || dx <= 0 || dout % dx != 0 || (y && (dy <= 0 || dout % dy != 0)))
        return fail(PGLAMD_E_SHAPE, "aggregate: dx=%lld dy=%lld dout=%lld is not a trailing-dim broadcast",
                    (long long)dx, (long long)dy, (long long)dout);
    if (reduce_op < 0 || reduce_op > 3 || (y && (message_op < 0 || message_op > 3)))
        return fail(PGLAMD_E_ARG, "aggregate: bad op enum");
    if (ex.x2 && (ex.x_split < 0 || ex.x_split >= INT32_MAX)) return fail(PGLAMD_E_RANGE, "aggregate_ext: x_split out of range");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define CALL(T) aggregate_typed<T>(x, dx, y, dy, eid, row, col, indptr, num_edges, n_csr_rows, out_rows, dout, \
                                   message_op, reduce_op, src_scale, dst_scale, accumulate, out, workspace, workspace_bytes, ex, st)
    switch (dtype) {
        case PGLAMD_F32: return CALL(float);
        case PGLAMD_F64: return CALL(double);
        case PGLAMD_I32: return CALL(int32_t);
        case PGLAMD_I64: return CALL(int64_t);
        case PGLAMD_F16: return CALL(__half);
        case PGLAMD_BF16: return CALL(__hip_bfloat16);
        default: return fail(PGLAMD_E_DTYPE, "aggregate: dtype %d not supported", dtype);
    }
#undef CALL
}

extern "C" int32_t pglamd_aggregate(const void* x, int32_t dtype, int64_t n_x_rows, int64_t dx, const void* y,
                                    int64_t dy, const int32_t* eid, const int32_t* row, const int32_t* col,
                                    const int64_t* indptr, int64_t num_edges, int64_t n_csr_rows,
                                    int64_t out_rows, int64_t dout, int32_t message_op, int32_t reduce_op,
                                    const float* src_scale, const float* dst_scale, int32_t accumulate, void* out,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    (void)n_x_rows;
    return aggregate_entry(x, dtype, dx, y, dy, eid, row, col, indptr, num_edges, n_csr_rows, out_rows, dout, message_op, reduce_op,
                           src_scale, dst_scale, accumulate, out, workspace, workspace_bytes, AggExtra{}, stream);
}

extern "C" int32_t pglamd_aggregate_ext(const void* x, const void* x2, int64_t x_split, int32_t dtype, int64_t dx, int64_t ldx,
                                        const void* y, int64_t dy, const int32_t* eid, const int32_t* row, const int32_t* col,
                                        const int64_t* indptr, const int64_t* zero_indptr, int64_t max_row_edges,
                                        int64_t num_edges, int64_t n_csr_rows, int64_t out_rows, int64_t dout, int64_t ldout,
                                        int32_t message_op, int32_t reduce_op, const float* dst_scale, int32_t accumulate,
                                        void* out, void* workspace, size_t workspace_bytes, int32_t flags, void* stream) {
    if (ldx < 0 || ldout < 0) return fail(PGLAMD_E_ARG, "aggregate_ext: negative row stride");
    if (flags & ~PGLAMD_AGG_DEAL_CHUNKS) return fail(PGLAMD_E_ARG, "aggregate_ext: unknown flag bits 0x%x", (unsigned)flags);
    AggExtra ex;
    ex.x2 = x2; ex.x_split = x_split; ex.zero_indptr = zero_indptr; ex.max_row_edges = max_row_edges; ex.ldx = ldx; ex.ldo = ldout;
    ex.flags = flags;
    return aggregate_entry(x, dtype, dx, y, dy, eid, row, col, indptr, num_edges, n_csr_rows, out_rows, dout, message_op, reduce_op,
                           nullptr, dst_scale, accumulate, out, workspace, workspace_bytes, ex, stream);
}

// ------------------------------------------------------------------------------------------------
// pglamd_aggregate_dense: aggregation feeding a dense layer inside one kernel (the flat kernel with SINK = 1)
// ------------------------------------------------------------------------------------------------
namespace pglamd {
__global__ __launch_bounds__(kBlock) void pack_weight_kernel(const float* __restrict__ w, int d_in, int d_out, float* __restrict__ wp) {
    // wp[(ct * (d_in / 4) + kk) * 64 + lane] = w[(4 kk + lane / 16) * d_out + ct * 16 + lane % 16]: the B operand of
    // v_mfma_f32_16x16x4_f32 for column tile ct and k-step kk, one coalesced 256-byte load per (ct, kk)
    const int KK = d_in / 4;
    const int64_t total = (int64_t)(d_out / 16) * KK * kWave;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int lane = (int)(i % kWave);
        const int64_t t = i / kWave;
        const int kk = (int)(t % KK), ct = (int)(t / KK);
        wp[i] = w[(int64_t)(4 * kk + lane / 16) * d_out + ct * 16 + (lane & 15)];
    }
}

template <int VEC, int SS = 0>
static int32_t launch_dense(AggParams p, hipStream_t st) {
    const int64_t nb = ceil_div(p.n_chunks, kWavesPerBlock);
    p.n_blocks = (int)nb;
    p.n_grid_chunks = (int)xcd_grid(nb);
    const int64_t zb = ceil_div(ceil_div(p.out_rows, kWave), kWavesPerBlock);
    const bool fixups = needs_fixups(p);
    if (fixups) PGLAMD_HIP_CHECK(hipMemsetAsync(p.long_count, 0, 2 * sizeof(int), st));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool profiling = prof().on.load(std::memory_order_relaxed);
    if (profiling) {
        { std::lock_guard<std::mutex> lk(prof().mu); prof().last_kernel = std::string("agg_flat_kernel<float, ") + (VEC == 2 ? "2" : "1") + ", 1, 0, 0, dense sink>"; }
        PGLAMD_HIP_CHECK(hipEventCreate(&e0));
        PGLAMD_HIP_CHECK(hipEventCreate(&e1));
        PGLAMD_HIP_CHECK(hipEventRecord(e0, st));
    }
    hipLaunchKernelGGL((agg_flat_kernel<float, VEC, 1, 0, 0, SS, true, 0, 1>), dim3((unsigned)(p.n_grid_chunks + zb)), dim3(kBlock), 0, st, p);
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

extern "C" size_t pglamd_aggregate_dense_workspace_bytes(int64_t num_edges, int64_t d_in, int64_t d_out) {
    return pglamd_aggregate_workspace_bytes(num_edges, d_in, PGLAMD_F32) + align_up((size_t)(d_in > 0 ? d_in : 1) * (size_t)(d_out > 0 ? d_out : 1) * 4, 256) + 256;
}

extern "C" int32_t pglamd_aggregate_dense(const float* x, int64_t d_in, const int32_t* row, const int32_t* col, const int64_t* indptr,
                                          int64_t num_edges, int64_t n_csr_rows, int64_t out_rows, int32_t reduce_op,
                                          const float* edge_scale, const float* dst_scale, const float* w, const float* bias, int32_t act,
                                          int64_t d_out, float* agg_out, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!out || !w || !indptr || (num_edges > 0 && (!x || !row || !col))) return fail(PGLAMD_E_ARG, "aggregate_dense: NULL pointer");
    if (num_edges < 0 || num_edges > kMaxEdges || n_csr_rows >= INT32_MAX || out_rows >= INT32_MAX)
        return fail(PGLAMD_E_RANGE, "aggregate_dense: sizes beyond int32 engine range");
    if ((d_in != 64 && d_in != 128) || d_out <= 0 || d_out % 16 != 0 || d_out > 1024)
        return fail(PGLAMD_E_SHAPE, "aggregate_dense: d_in must be 64 or 128 and d_out a multiple of 16 up to 1024 (got %lld, %lld)", (long long)d_in, (long long)d_out);
    if (reduce_op != PGLAMD_SUM && reduce_op != PGLAMD_MEAN) return fail(PGLAMD_E_ARG, "aggregate_dense: sum or mean");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(agg_out) | reinterpret_cast<uintptr_t>(workspace)) % 8 != 0)
        return fail(PGLAMD_E_ARG, "aggregate_dense: x / agg_out / workspace must be 8-byte aligned");
    if (!workspace || workspace_bytes < pglamd_aggregate_dense_workspace_bytes(num_edges, d_in, d_out))
        return fail(PGLAMD_E_WORKSPACE, "aggregate_dense: workspace too small");
    if (out_rows == 0) return PGLAMD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // packed weight at the head of the workspace, the aggregation's own partial buffers behind it
    float* wp = static_cast<float*>(workspace);
    const size_t wp_bytes = align_up((size_t)d_in * (size_t)d_out * 4, 256);
    // two forms (aggregate_dense2.hpp says why): the specialised workgroup (W resident in LDS, a matrix wave) where W fits,
    // the per-wave tiles of the flat kernel (SINK = 1) otherwise.  PGLAMD_DENSE_FORM=1 | 2 forces one (A/B runs).
    static const int form_env = [] { const char* e = getenv("PGLAMD_DENSE_FORM"); return e ? atoi(e) : 0; }();
    const bool form2 = form_env != 1 && dense2_covers(d_in, d_out) && reinterpret_cast<uintptr_t>(w) % 16 == 0;
    if (form_env == 2 && !form2) return fail(PGLAMD_E_SHAPE, "aggregate_dense: form 2 needs d_in * d_out * 4 + ring <= 80 KB");
    {
        const int64_t total = form2 ? d_in * d_out : (d_out / 16) * (d_in / 4) * kWave;
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(total, kBlock), 1024);
        if (!form2) {                                         // (form 2 packs W while it stages it into LDS)
            hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(kBlock), 0, st, w, (int)d_in, (int)d_out, wp);
            PGLAMD_LAUNCH_CHECK();
        }
    }
    char* ws = static_cast<char*>(workspace) + wp_bytes;
    AggParams p{};
    p.x = x; p.x2 = x; p.x_split = INT32_MAX; p.out = agg_out; p.row = row; p.col = col; p.indptr = indptr; p.zero_indptr = indptr;
    p.dst_scale = dst_scale;
    p.src_scale = edge_scale; p.ss_by_pos = edge_scale ? 1 : 0;      // one scale per edge POSITION of the sorted stream
    p.ldx = d_in; p.ldo = d_in; p.out_rows = out_rows; p.n_csr_rows = n_csr_rows; p.E = (int)num_edges;
    p.is_mean = reduce_op == PGLAMD_MEAN; p.gy = 1;
    p.w = w; p.wp = wp; p.bias = bias; p.out2 = out; p.dout2 = (int)d_out; p.act = act;
    p.j_base = 0; p.tile_cols = (int)d_in; p.zvec = 2; p.align = 1;
    const int K = chunk_edges_for(num_edges);
    p.chunk = K;
    p.n_chunks = (int)ceil_div(num_edges > 0 ? num_edges : 1, (int64_t)K);
    const size_t half = align_up((size_t)p.n_chunks * d_in * sizeof(float), 256);
    const size_t lst = align_up((size_t)(p.n_chunks + 64) * sizeof(int), 256);
    p.part_head = ws;
    p.part_tail = ws + half;
    p.long_count = reinterpret_cast<int*>(ws + 2 * half);
    p.long_list = p.long_count + 64;
    p.long_list2 = reinterpret_cast<int*>(ws + 2 * half + lst);
    if (num_edges == 0) p.n_chunks = 0;                      // only the empty-row roles run
    if (edge_scale) {
        if (form2) return d_in == 128 ? launch_dense2<2, true>(p, st) : launch_dense2<1, true>(p, st);
        return d_in == 128 ? launch_dense<2, 2>(p, st) : launch_dense<1, 2>(p, st);
    }
    if (form2) return d_in == 128 ? launch_dense2<2>(p, st) : launch_dense2<1>(p, st);
    return d_in == 128 ? launch_dense<2>(p, st) : launch_dense<1>(p, st);
}

extern "C" int32_t pglamd_profile_begin(void) {
    std::lock_guard<std::mutex> lk(prof().mu);
    for (auto& pr : prof().ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    prof().ev.clear();
    prof().on = true;
    return PGLAMD_OK;
}

extern "C" int32_t pglamd_profile_end(double* total_ms, int64_t* launches) {
    prof().on = false;
    std::lock_guard<std::mutex> lk(prof().mu);
    double tot = 0;
    int64_t n = 0;
    for (auto& pr : prof().ev) {
        float ms = 0;
        PGLAMD_HIP_CHECK(hipEventSynchronize(pr.second));
        PGLAMD_HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms; ++n;
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    prof().ev.clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return PGLAMD_OK;
}

extern "C" const char* pglamd_profile_last_kernel(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(prof().mu);
    copy = prof().last_kernel;
    return copy.c_str();
}
