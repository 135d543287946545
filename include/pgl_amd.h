/*
 * pgl_amd.h -- C ABI of libpglamd.so: the MI355X (gfx950) message-passing engine that sits
 * behind PGL's Graph.send / recv / send_recv / send_ue_recv / send_uv and pgl.math.segment_*.
 *
 * The reference (PaddlePaddle/PGL 2.2.6) has no plugin registry for this path: the seam is the
 * set of paddle.* tensor ops its Python makes.  Each entry point below
 * replaces one of those call sites; the reference file:line it stands in for is cited on it.
 * INTEGRATION.md shows the ctypes stub a PGL maintainer would add on the reference side.
 *
 * Conventions (all entry points)
 *   - plain pointers + sizes; every pointer is a DEVICE pointer unless it says "host";
 *     no torch / paddle types anywhere in a signature;
 *   - row-major, rows contiguous (last dim fastest), same as paddle/torch default layout;
 *   - the caller owns every buffer, including outputs and scratch: query
 *     pglamd_<op>_workspace_bytes(...) first, allocate, pass `workspace`;
 *     the library never allocates or frees device memory on these paths;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t, NULL = default stream);
 *     no hidden synchronisation unless the entry point says so;
 *   - return 0 on success, a negative PGLAMD_E_* otherwise; pglamd_last_error() gives the
 *     thread-local message;
 *   - index validity (ids < num_nodes) is the caller's contract, exactly as in the reference.
 *
 * "CSR order" below means the order graph_kernel.build_index emits (pgl/graph_kernel.pyx:59-88):
 * edges stably sorted by key u (= dst for adj_dst_index, pgl/graph.py:1319-1328), ascending
 * original edge id within a row.
 */
#ifndef PGL_AMD_H_
#define PGL_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGLAMD_ABI_VERSION 4

/* status codes */
#define PGLAMD_OK 0
#define PGLAMD_E_SHAPE (-1)     /* bad size / unsupported broadcast */
#define PGLAMD_E_DTYPE (-2)     /* dtype not supported by this entry point */
#define PGLAMD_E_RANGE (-3)     /* num_nodes / num_edges beyond the int32 engine range */
#define PGLAMD_E_WORKSPACE (-4) /* workspace NULL or too small */
#define PGLAMD_E_HIP (-5)       /* HIP runtime error (message has hipGetErrorString) */
#define PGLAMD_E_ARG (-6)       /* NULL pointer / bad enum */
#define PGLAMD_E_UNAVAILABLE (-7) /* optional helper library (METIS, RCCL) not present */
#define PGLAMD_E_RCCL (-8)      /* RCCL error (message has ncclGetErrorString) */

/* element types */
#define PGLAMD_F16 0
#define PGLAMD_F32 1
#define PGLAMD_F64 2
#define PGLAMD_I32 3
#define PGLAMD_I64 4
#define PGLAMD_BF16 5

/* reduce_op of send_u_recv / send_ue_recv / segment_* ("sum","mean","max","min") */
#define PGLAMD_SUM 0
#define PGLAMD_MEAN 1
#define PGLAMD_MAX 2
#define PGLAMD_MIN 3

/* message_op of send_ue_recv / send_uv ("add","sub","mul","div") */
#define PGLAMD_ADD 0
#define PGLAMD_SUB 1
#define PGLAMD_MUL 2
#define PGLAMD_DIV 3

int32_t pglamd_abi_version(void);
/* Process-wide TUNING option (tests and experiments; every hot-path choice a caller makes is a per-call argument -- see
 * PGLAMD_AGG_DEAL_CHUNKS -- and the entry points are re-entrant).
 * "csr_onesweep": how pglamd_csr_build sorts.  -1 (default) = by size: one-sweep passes (one histogram of all digits, then ONE kernel per
 * digit with decoupled look-back: 7 launches) for up to 1 M edges, the multi-kernel passes (histogram / scan / scatter per digit: 12
 * launches, XCD-local tile order) above; 0 = always multi-kernel; g >= 1 = always one-sweep with g consecutive tiles per XCD.  The output
 * is bit-identical either way. */
int32_t pglamd_set_option(const char* name, int64_t value);
const char* pglamd_last_error(void);
/* name of the device the library sees, e.g. "gfx950..." (host string, valid until next call) */
const char* pglamd_device_arch(void);

/* ------------------------------------------------------------------------------------------------
 * K8  CSR build.   Replaces graph_kernel.build_index (pgl/graph_kernel.pyx:59-88) as called by
 * EdgeIndex.from_edges (pgl/utils/edge_index.py:38-58; numpy path :56-57, tensor path :42-54).
 *   u, v            int64 keys / neighbours, element strides u_stride / v_stride (so a column of
 *                   the [E,2] edge array can be passed in place: stride 2)
 *   degree[N] sorted_v[E] sorted_u[E] sorted_eid[E] indptr[N+1]     int64, reference layout,
 *                   bit-identical to build_index (stable by original edge id)
 *   row32[E] col32[E] eid32[E]   optional (NULL to skip) int32 copies of sorted_u / sorted_v /
 *                   sorted_eid that the aggregation kernels read (halves index traffic)
 *   range_flag      optional device int32[1] (caller zeroes it): set to 1 when a key lies outside
 *                   [0, N) or a neighbour id outside [0, 2^31).  Offending keys are clamped to row 0
 *                   so the call itself stays memory-safe; the call is asynchronous, the CALLER
 *                   decides when to read the flag (pgl_amd.ops.csr_build raises ValueError, the
 *                   error pglamd_build_index_host returns as PGLAMD_E_RANGE on the host side).
 * Limits: N < 2^31, E < 2^31 (PGLAMD_E_RANGE otherwise).
 * ---------------------------------------------------------------------------------------------- */
size_t pglamd_csr_build_workspace_bytes(int64_t num_edges, int64_t num_nodes);
int32_t pglamd_csr_build(const int64_t* u, int64_t u_stride, const int64_t* v, int64_t v_stride,
                         int64_t num_edges, int64_t num_nodes, int64_t* degree, int64_t* sorted_v,
                         int64_t* sorted_u, int64_t* sorted_eid, int64_t* indptr, int32_t* row32,
                         int32_t* col32, int32_t* eid32, int32_t* range_flag, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Replaces unique_segment(dst_sorted) = paddle.unique(return_inverse=True)
 * (pgl/utils/helper.py:156-160, cached by Graph.get_segment_ids, pgl/graph.py:1397-1407).
 *   uniq_ind[<=N]   ascending ids of rows with degree > 0 (first *num_uniq entries valid)
 *   segment_ids[E]  dense rank of each CSR-ordered edge's row
 *   num_uniq        device int64[1]                                                              */
size_t pglamd_unique_segment_workspace_bytes(int64_t num_edges, int64_t num_nodes);
int32_t pglamd_unique_segment(const int64_t* degree, const int64_t* sorted_u, int64_t num_edges,
                              int64_t num_nodes, int64_t* uniq_ind, int64_t* segment_ids,
                              int64_t* num_uniq, void* workspace, size_t workspace_bytes,
                              void* stream);

/* strided int64 -> int32 narrowing copy (src/dst columns of the edge array for send_uv) */
int32_t pglamd_narrow_i64(const int64_t* in, int64_t in_stride, int64_t n, int32_t* out,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * K1 / K2  aggregation over a CSR-ordered edge list (atomic-free, deterministic).
 * Replaces paddle.geometric.send_u_recv  (pgl/graph.py:859-861, 885-887: Graph.send_recv /
 * send_u_recv) and paddle.geometric.send_ue_recv (pgl/graph.py:929-937: Graph.send_ue_recv):
 *
 *     out[r, j] = dst_scale[r] * REDUCE_{p : row[p]==r}  ( src_scale[col[p]] * x[col[p], xj] (mop) y[yp, yj] )
 *
 *   x          [n_x_rows, dx]   node features, `dtype`
 *   row, col   [E] int32, CSR order (row non-decreasing): destination row / source row per edge
 *   indptr     [n_csr_rows+1] int64 (the reference's indptr)
 *   y          NULL for send_u_recv; else edge features [E, dy] of `dtype` in ORIGINAL edge order
 *              when eid != NULL (yp = eid[p]) or already in CSR order when eid == NULL (yp = p).
 *              Routing note: y = [E, 1] (dy == 1) with eid == NULL, message_op MUL, reduce SUM / MEAN, F32 and rows wider
 *              than 128 bytes is one fp32 value per edge POSITION multiplied into the gathered row -- the library answers it
 *              with the flat kernel's per-position scale slot (the `edge_scale` of pglamd_aggregate_dense: 4 sequential
 *              bytes per edge) instead of the general edge-operand path.  Same result up to the order of one multiplication
 *              (x * y summed, both ways); every accumulate mode and out_rows < n_csr_rows behave as documented below
 *              (tests/test_a7_a9_attention_ops.py::test_abi_edge_operand_e1_mul_reroute_equals_the_general_path).
 *   dx, dy, dout   trailing sizes.  Broadcast rule supported on the fast path: trailing-dim
 *              broadcast only, xj = j / (dout/dx), yj = j / (dout/dy)  (covers [H,D]x[H,1],
 *              [H,D]x[H,D], [D]x[1]); other numpy patterns must be expanded by the caller.
 *   src_scale, dst_scale   optional float32 [n_x_rows] / [out_rows] (NULL = 1): fuses GCN's
 *              symmetric degree normalisation (pgl/nn/conv.py:242,250) into the aggregation;
 *              only for floating dtypes with reduce_op SUM / MEAN.
 *   out        [out_rows, dout]; every row is written exactly once (rows without messages = 0,
 *              as the reference guarantees); out_rows may exceed n_csr_rows (out_size semantics)
 *   reduce_op  PGLAMD_SUM/MEAN/MAX/MIN ; message_op PGLAMD_ADD/SUB/MUL/DIV (ignored if y NULL)
 *   accumulate 0: out is overwritten (rows without edges = 0).  1: rows that receive edges are
 *              combined (+ / max / min) with their existing contents, other rows are untouched --
 *              used to add the halo-source edges after the local-source edges of a partitioned
 *              graph (still deterministic: the two launches are ordered on the stream); not with MEAN.
 *              2: rows that receive edges are OVERWRITTEN, other rows are untouched -- the boundary rows of a
 *              partitioned graph are finished after the halo arrives, on top of the interior rows' launch
 *              (max / min have no identity that an empty first launch could leave behind).
 * ---------------------------------------------------------------------------------------------- */
size_t pglamd_aggregate_workspace_bytes(int64_t num_edges, int64_t dout, int32_t dtype);
int32_t pglamd_aggregate(const void* x, int32_t dtype, int64_t n_x_rows, int64_t dx, const void* y,
                         int64_t dy, const int32_t* eid, const int32_t* row, const int32_t* col,
                         const int64_t* indptr, int64_t num_edges, int64_t n_csr_rows,
                         int64_t out_rows, int64_t dout, int32_t message_op, int32_t reduce_op,
                         const float* src_scale, const float* dst_scale, int32_t accumulate, void* out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* K1x  pglamd_aggregate with what a ROW-PARTITIONED graph needs (pgl_amd/distributed.py; the reference's multi-GPU graph,
 * pgl/graph.py:1475-1553, all-reduces [N, d] partial sums instead).  Every argument shared with pglamd_aggregate means the same;
 * there is no src_scale (a caller scales owned rows before they travel).  The additions:
 *   x2, x_split     second source table: a column id c >= x_split reads row (c - x_split) of x2, c < x_split reads row c of x.
 *                   x = the rows this rank owns, x2 = the rows received from its peers: the boundary rows of a partition
 *                   aggregate straight from the two buffers, nothing is copied into an [owned | halo] matrix first.
 *                   x2 NULL: one table, as pglamd_aggregate.  Same dtype and row length (dx == dout) for both.
 *   zero_indptr     [n_csr_rows+1] or NULL (= indptr).  With accumulate 0 a row r is zero-filled iff
 *                   zero_indptr[r] == zero_indptr[r+1]; rows that are empty in THIS index but not in zero_indptr are left
 *                   untouched.  The interior launch of a partition passes the indptr of ALL local edges: it writes interior
 *                   rows and truly empty rows, the boundary launch (accumulate 2) writes the rest -- every output row is
 *                   written exactly once across the two launches.
 *   max_row_edges   longest row of the index, or 0 when unknown.  No row of <= chunk edges is ever split between waves, so a
 *                   launcher that sees max_row_edges <= its chunk skips the counter reset and the fix-up launch
 *                   (pack indices of a halo plan: every row has one edge).
 *   ldx, ldout      row strides in elements of x (and x2) and of out; 0 = dense (dx / dout).  A launch may read and write a
 *                   COLUMN BLOCK of wider matrices: pass the address of the block's first column, its width as dx / dout and
 *                   the full row length as the stride.  The column-pipelined halo exchange aggregates columns [0, d/2) of the
 *                   received rows while columns [d/2, d) are still on the wire.  (With accumulate 0 a strided output needs
 *                   num_edges > 0: the stand-alone zero-fill of an edgeless index is dense.)
 *   flags           per-call launch choices (ABI 4; replaces the process-wide "xcd_swizzle" option of ABI <= 3, which two threads
 *                   with different plans raced on).  PGLAMD_AGG_DEAL_CHUNKS: the chunks of the destination-sorted edge stream are
 *                   dealt round the 8 XCDs instead of running in contiguous blocks per XCD (the default: neighbouring chunks share
 *                   an L2, which partition- / cluster-ordered graphs use) -- for row orders that correlate with row LENGTH
 *                   (pgl_amd.distributed.HaloPlan(row_order="peers")), where the blocked mapping gives one XCD all the short rows.
 * ---------------------------------------------------------------------------------------------- */
#define PGLAMD_AGG_DEAL_CHUNKS 1
int32_t pglamd_aggregate_ext(const void* x, const void* x2, int64_t x_split, int32_t dtype, int64_t dx,
                             int64_t ldx, const void* y, int64_t dy, const int32_t* eid, const int32_t* row,
                             const int32_t* col, const int64_t* indptr, const int64_t* zero_indptr,
                             int64_t max_row_edges, int64_t num_edges, int64_t n_csr_rows,
                             int64_t out_rows, int64_t dout, int64_t ldout, int32_t message_op, int32_t reduce_op,
                             const float* dst_scale, int32_t accumulate, void* out, void* workspace,
                             size_t workspace_bytes, int32_t flags, void* stream);

/* K1d  aggregation feeding a dense layer inside ONE kernel (row f1: "SpMM -> GEMM epilogue"; GCNConv's
 * send_recv(sum) -> linear -> + bias -> activation, pgl/nn/conv.py:242-254, when input_size <= output_size):
 *     agg[r, :] = dst_scale[r] * REDUCE_{p : row[p]==r} x[col[p], :]            (SUM or MEAN, F32, d_in = 64 or 128)
 *     out[r, :] = act( agg[r, :] @ w + bias )                                    (w [d_in, d_out] row-major, d_out % 16 == 0)
 * Two forms behind this entry (pgl_amd/csrc/aggregate_dense2.hpp says why):
 *   - where w fits in LDS (d_in * d_out * 4 + ring <= 80 KB): persistent workgroups of 8 producer waves + 4 matrix waves; the
 *     producers walk the edges and hand finished rows through a ring in LDS to the matrix waves, which hold w in LDS for the
 *     life of the workgroup and multiply 16 rows at a time with v_mfma_f32_16x16x4_f32;
 *   - otherwise: every wave of the flat aggregation kernel parks its finished rows in a 16-row LDS tile of its own and
 *     multiplies it by w (B operand from memory).
 * fp32 inputs, fp32 accumulation -- the reference's arithmetic up to re-association; only `out` is written: the [n_rows, d_in]
 * intermediate never travels through HBM, and the matrix cores run in the shadow of the row gathers.
 * edge_scale (optional, ABI 2): one fp32 value per edge POSITION of the sorted stream, multiplied into the gathered row
 *     (agg[r] = dst_scale[r] * SUM_p edge_scale[p] * x[col[p]]): GCN's source-side degree norm (pgl/nn/conv.py:242) laid out
 *     along the stream once per graph -- 4 sequential bytes per edge instead of a pass over [N, d_in] per layer.
 * agg_out (optional): the aggregated rows are ALSO stored (training: the weight gradient is agg^T @ d out).
 * act: 0 none, 1 relu.  Rows without edges get act(bias).  Rows longer than a chunk go through the split-row fix-up and get the
 * layer from a small MFMA kernel afterwards.  Deterministic.  w 16-byte aligned for the first form.  Workspace (8-byte aligned):
 * pglamd_aggregate_dense_workspace_bytes. */
size_t pglamd_aggregate_dense_workspace_bytes(int64_t num_edges, int64_t d_in, int64_t d_out);
int32_t pglamd_aggregate_dense(const float* x, int64_t d_in, const int32_t* row, const int32_t* col,
                               const int64_t* indptr, int64_t num_edges, int64_t n_csr_rows,
                               int64_t out_rows, int32_t reduce_op, const float* edge_scale,
                               const float* dst_scale, const float* w, const float* bias, int32_t act, int64_t d_out,
                               float* agg_out, float* out, void* workspace, size_t workspace_bytes,
                               void* stream);

/* Gradients that pgl_amd/autograd.py used to compose from [E, d] row gathers (F32; PaddlePaddle implements them inside
 * graph_send_recv_grad / graph_send_ue_recv_grad behind the call sites at pgl/graph.py:834-937).
 *
 * pglamd_winner_grad: d x of send_recv(x, max | min).  The gradient of an output row goes to EVERY message equal to the winner
 * (Paddle's rule):  grad_x[u, j] = sum_{e = (u -> v)} [x[u, j] == out[v, j]] * grad_out[v, j].  One walk of the SRC-sorted
 * stream (src_row / src_col / src_indptr = the src-keyed CSR: row = u, col = v), d <= 256, deterministic.
 *
 * pglamd_edge_operand_grad: d y of send_ue_recv(x, y, message_op, sum | mean) for trailing-dim broadcast operands
 * (y [E, dy], d % dy == 0; the [E, H, 1] case also has pglamd_sddmm):
 *     grad_y[eid[p], jy] = sum_{j in group jy} dst_scale[row[p]] * grad_out[row[p], j] * { 1 | -1 | x[col[p], j] | -x[col[p], j] / y^2 }
 * for ADD | SUB | MUL | DIV; row / col / eid = the dst-sorted CSR (eid NULL: grad_y in CSR order); dst_scale NULL = 1
 * (MEAN: 1 / in-degree); x is needed for MUL / DIV, y (original edge order, like grad_y) for DIV. */
size_t pglamd_winner_grad_workspace_bytes(int64_t num_edges, int64_t d);
int32_t pglamd_winner_grad(const float* grad_out, const float* out, const float* x, int64_t d,
                           const int32_t* src_row, const int32_t* src_col, const int64_t* src_indptr,
                           int64_t num_edges, int64_t n_x_rows, float* grad_x, void* workspace,
                           size_t workspace_bytes, void* stream);
int32_t pglamd_edge_operand_grad(const float* grad_out, const float* x, const float* y,
                                 const float* dst_scale, int64_t d, int64_t dy, const int32_t* row,
                                 const int32_t* col, const int32_t* eid, int64_t num_edges,
                                 int32_t message_op, float* grad_y, void* stream);

/* Measurement hook for the dominant kernel (bench.py roofline leg): between profile_begin and
 * profile_end every launch of the flat aggregation kernel is bracketed by HIP events on its own
 * launch stream; profile_end synchronises them and returns the summed kernel time (host out). */
int32_t pglamd_profile_begin(void);
int32_t pglamd_profile_end(double* total_ms, int64_t* launches);
/* name + template arguments of the flat kernel the last pglamd_aggregate call launched (host string) */
const char* pglamd_profile_last_kernel(void);

/* K1'  un-indexed COO variant (edges in arbitrary order, hardware float atomics; SUM only, F32): what answers
 * paddle.geometric.send_u_recv(x, src, dst) (pgl/graph.py:859-861; Paddle-free fallback pgl/utils/helper.py:163-210) on a SMALL
 * edge list used once, where the launches of a CSR build cost more than the aggregation: |E| * d <= 6.4 M elements (measured on
 * MI355X, profiles/r05/coo.txt: 0.023 vs 0.074 ms at 13 k edges, d = 128; break-even near 50 k edges).  Above that the memory-side
 * read-modify-write of every 512-byte destination row makes it 3-7 x slower than pglamd_csr_build + pglamd_aggregate, which is
 * what pgl_amd.ops.send_u_recv runs there.  Result is order-nondeterministic in the last bits, unlike pglamd_aggregate.  `out`
 * is zero-filled here. */
int32_t pglamd_scatter_add_coo(const float* x, int64_t d, const int32_t* src, const int32_t* dst,
                               int64_t num_edges, int64_t out_rows, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3  send_uv.  Replaces paddle.geometric.send_uv (pgl/graph.py:964-966):
 *     out[e, j] = x[src[e], j/(dout/dx)] (mop) y[dst[e], j/(dout/dy)]      (original edge order)
 * ---------------------------------------------------------------------------------------------- */
int32_t pglamd_send_uv(const void* x, const void* y, int32_t dtype, int64_t dx, int64_t dy,
                       int64_t dout, const int32_t* src, const int32_t* dst, int64_t num_edges,
                       int32_t message_op, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K5  segment reduce.  Replaces paddle.geometric.segment_{sum,mean,max,min}
 * (pgl/math.py:30-178; Message.reduce_*, pgl/message.py:34-105).
 *   data [E, d] of `dtype`; ids [E] sorted non-decreasing, int32 (ids_i64 = 0) or int64 (= 1);
 *   out [n_out_rows, d] with n_out_rows = ids[E-1] + 1 supplied by the caller; rows whose id
 *   never occurs are 0.
 * ---------------------------------------------------------------------------------------------- */
size_t pglamd_segment_reduce_workspace_bytes(int64_t num_rows, int64_t d, int64_t n_out_rows,
                                             int32_t dtype);
int32_t pglamd_segment_reduce(const void* data, int32_t dtype, const void* ids, int32_t ids_i64,
                              int64_t num_rows, int64_t d, int64_t n_out_rows, int32_t reduce_op,
                              void* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4  segment softmax.  Replaces pgl.math.segment_softmax (pgl/math.py:181-224: segment_max,
 * gather, sub, exp, segment_sum, gather, div) and -- with perm32 = sorted_eid -- the whole of
 * GF.edge_softmax (pgl/nn/functional/graph_op.py:101-123: gather by eid, segment_softmax,
 * scatter back by eid): data and out are only ever addressed in the data's OWN order.
 *   data/out        [num_rows, d] F32 or F64 (out != data)
 *   row32           [num_rows] segment id of the p-th element in SEGMENT-SORTED order (non-decreasing)
 *   perm32          [num_rows] position in `data` of the p-th sorted element, or NULL if data is
 *                   already sorted by segment (pgl.math.segment_softmax)
 *   seg_of_elem32   [num_rows] segment id of data row i (== row32 when perm32 is NULL; the dst
 *                   column of the edge list for edge_softmax by dst)
 *   seg_ptr         [n_seg+1] int64 offsets of each segment in sorted order (the reference's indptr)
 * Arithmetic is the reference's: e = exp(x - max_seg); out = e / sum_seg(e).  Deterministic.
 * ---------------------------------------------------------------------------------------------- */
size_t pglamd_segment_softmax_workspace_bytes(int64_t num_rows, int64_t d, int64_t n_seg, int32_t dtype);
int32_t pglamd_segment_softmax(const void* data, int32_t dtype, int64_t num_rows, int64_t d,
                               const int32_t* row32, const int32_t* perm32,
                               const int32_t* seg_of_elem32, const int64_t* seg_ptr, int64_t n_seg,
                               void* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3+K4+K2 fused: the attention aggregation of GATConv (pgl/nn/conv.py:331-339) in ONE pass over
 * the dst-sorted edges, with an online softmax per (row, head):
 *     out[v,h,:] = sum_{e=(u->v)} softmax_v( leaky_relu(attn_src[u,h] + attn_dst[v,h], slope) ) * feature[u,h,:]
 * == send_uv("add") -> leaky_relu -> edge_softmax(by dst) -> send_ue_recv("mul","sum") of the
 * reference, without materialising any [E,H] tensor.  F32.  heads*head_dim <= 256.
 *   feature [N, heads, head_dim], attn_src/attn_dst [N, heads]; row/col/indptr: dst-sorted CSR
 *   out [out_rows, heads, head_dim] (rows without in-edges = 0)
 *   row_max,row_sum  optional [out_rows, heads]: the softmax statistics (max logit, sum of
 *                    exp(logit - max)) from which a backward pass can recompute alpha per edge.
 * Deterministic (no atomics).  Softmax sums are accumulated with rescaling, so alpha differs from
 * the reference's two-pass evaluation only by fp32 rounding (tests: <= 1e-5 relative on out).
 * ---------------------------------------------------------------------------------------------- */
size_t pglamd_gat_aggregate_workspace_bytes(int64_t num_edges, int64_t heads, int64_t head_dim);
/* drop_p in [0,1): attention dropout applied to alpha (pgl/nn/conv.py:337-338) as a counter-based
 * hash of (seed, original edge id, head); needs eid (dst-sorted sorted_eid) when drop_p > 0.  *   out_pos, sum_pos   optional (both or neither; training): [out_rows, H*D] and [out_rows, H] positive-part statistics
 *                      consumed by pglamd_gat_backward (see there).
 */
int32_t pglamd_gat_aggregate(const float* feature, const float* attn_src, const float* attn_dst,
                             int64_t heads, int64_t head_dim, float negative_slope, float drop_p,
                             uint32_t seed, const int32_t* row, const int32_t* col,
                             const int32_t* eid, const int64_t* indptr, int64_t num_edges,
                             int64_t n_csr_rows, int64_t out_rows, float* out, float* row_max,
                             float* row_sum, float* out_pos, float* sum_pos, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Backward of pglamd_gat_aggregate (row "next" f1: fused backward), alpha recomputed per edge from
 * the forward's statistics; nothing of size [E,H] or [E,H,D] is materialised:
 *   grad_feature[u,h,:] = sum_{e=(u->v)} drop_e alpha_e grad_out[v,h,:]
 *   grad_attn_src[u,h]  = sum_{e=(u->v)} d pre_e          (both from ONE walk of the src-sorted CSR)
 *   grad_attn_dst[v,h]  = sum_{e=(u->v)} d pre_e          (one walk of the dst-sorted CSR)
 *   grad_pre (optional) : when non-NULL the src-sorted walk also writes d pre_e to grad_pre[E,H], row p = the p-th edge of
 *                         the SRC-SORTED stream (src_row / src_col), and the dst-sorted walk is skipped; the caller then
 *                         takes grad_attn_dst as the segment sum of grad_pre by destination (pglamd_aggregate over the
 *                         dst-sorted CSR with col = the src-sorted position of each edge): 1.3 ms faster at C3 for
 *                         4*E*H bytes of scratch.  grad_attn_dst may be NULL in that case.
 *   out_pos, sum_pos (optional, from pglamd_gat_aggregate): the forward's positive-part statistics -- out_pos[v,h,:] = the part
 *                         of out[v] contributed by edges with pre_e > 0, sum_pos[v,h] = their softmax mass.  With them
 *                         grad_attn_dst[v,h] = <g,out_pos> + slope <g,out - out_pos> - t (sum_pos + slope (1 - sum_pos)) is
 *                         evaluated per (node, head) inside the pack kernel: no per-edge buffer (grad_pre is ignored) and
 *                         no dst-sorted walk -- the round-2 default of pgl_amd (0.8 ms less at C3 for one more [N,H*D]
 *                         tensor kept from the forward).
 *   d pre_e = alpha_e (drop_e <grad_out[v,h,:], feature[u,h,:]> - t[v,h]) * leaky_relu'(attn_src[u,h] + attn_dst[v,h])
 *   t[v,h]  = sum_d grad_out[v,h,d] * out[v,h,d]  (out = the forward's output; computed here).
 *   dst_* / src_*  the dst-sorted and src-sorted CSRs (int32 row / col / eid, int64 indptr).
 * Replaces the backward of the four-op composition at pgl/nn/conv.py:331-339 (send_uv -> leaky_relu ->
 * edge_softmax -> send_ue_recv).  Workspace (256-byte aligned): pglamd_gat_backward_workspace_bytes.  Same shape
 * limits as the forward, and head_dim / VEC must be a power of two (per-head dot products are shuffle reductions). */
size_t pglamd_gat_backward_workspace_bytes(int64_t num_edges, int64_t num_nodes, int64_t heads,
                                           int64_t head_dim);
int32_t pglamd_gat_backward(const float* grad_out, const float* feature, const float* attn_src,
                            const float* attn_dst, const float* row_max, const float* row_sum,
                            const float* out, int64_t heads, int64_t head_dim, float negative_slope,
                            float drop_p, uint32_t seed, const int32_t* dst_row,
                            const int32_t* dst_col, const int32_t* dst_eid, const int64_t* dst_indptr,
                            const int32_t* src_row, const int32_t* src_col, const int32_t* src_eid,
                            const int64_t* src_indptr, int64_t num_edges, int64_t num_nodes,
                            float* grad_feature, float* grad_attn_src, float* grad_attn_dst,
                            float* grad_pre, const float* out_pos, const float* sum_pos,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Additive attention score over a sorted edge stream (GATv2Conv, pgl/nn/conv.py:421-424: send_uv(f, f, "add") ->
 * leaky_relu -> (alpha * attn).sum(-1), which materialises two [E,H,D] tensors):
 *   out[eid[p], h] = sum_d w[h,d] * leaky_relu( x_by_col[col[p],h,d] + y_by_row[row[p],h,d] )      (eid NULL: out[p,h])
 * F32, heads*head_dim <= 256, head_dim/VEC a power of two. */
int32_t pglamd_add_score(const float* x_by_col, const float* y_by_row, const float* w, int64_t heads,
                         int64_t head_dim, float negative_slope, const int32_t* row, const int32_t* col,
                         const int32_t* eid, int64_t num_edges, float* out, void* stream);

/* Backward of pglamd_add_score w.r.t. the ROW node's operand and (optionally) w, one walk of the sorted stream:
 *   grad_rows[r,h,d]          = sum_{p in row r} grad_score[gi_p,h] * w[h,d] * leaky_relu'(x_by_col[col_p] + y_by_row[r])
 *   grad_w_partials[c, h*D+d] = sum_{p in chunk c} grad_score[gi_p,h] * leaky_relu(x_by_col[col_p] + y_by_row[r])
 * with gi_p = eid[p] (p when eid is NULL) and c over pglamd_add_score_chunks(num_edges) chunks (the caller sums the
 * partials over c; NULL skips them).  The gradient w.r.t. the COLUMN node's operand is the same call on the transposed
 * index with the two operands swapped.  Workspace: pglamd_gat_aggregate_workspace_bytes. */
int64_t pglamd_add_score_chunks(int64_t num_edges);
int32_t pglamd_add_score_backward(const float* x_by_col, const float* y_by_row, const float* w,
                                  const float* grad_score, int64_t heads, int64_t head_dim,
                                  float negative_slope, const int32_t* row, const int32_t* col,
                                  const int32_t* eid, const int64_t* indptr, int64_t num_edges,
                                  int64_t num_rows, float* grad_rows, float* grad_w_partials,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* SDDMM over a sorted edge stream:  out[eid[p], h] = < x_by_col[col[p], h, :], y_by_row[row[p], h, :] >
 * (eid NULL: out[p, h]).  With (row, col, eid) = the dst-sorted CSR, x = node features and
 * y = the incoming gradient this is d loss / d edge_feature of Graph.send_ue_recv(x, e, "mul", "sum")
 * for e of shape [E, H, 1] (pgl/graph.py:929-937 backward) without materialising [E,H,D].  F32,
 * heads*head_dim <= 256, head_dim/VEC a power of two. */
int32_t pglamd_sddmm(const float* x_by_col, const float* y_by_row, int64_t heads, int64_t head_dim,
                     const int32_t* row, const int32_t* col, const int32_t* eid, int64_t num_edges,
                     float* out, void* stream);

/* out[i] = in[0] + ... + in[i-1] (int64; out may alias in).  The prefix sums the reference takes with paddle.cumsum
 * (pgl/utils/edge_index.py:54 indptr from degrees; the offsets of a sampled block's neighbour lists, pgl/sampling/sage.py:144-147)
 * as three small in-tree kernels (csrc/scan.hpp) -- no library primitive sits on the mini-batch path. */
size_t pglamd_exclusive_scan_i64_workspace_bytes(int64_t n);
int32_t pglamd_exclusive_scan_i64(const int64_t* in, int64_t n, int64_t* out, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* K4'  segment boundaries from sorted ids: seg_ptr[n_seg+1] (int64), n_seg = ids[E-1]+1 given by
 * the caller.  Used when segment_softmax is called with raw ids (pgl.math API). */
int32_t pglamd_seg_ptr_from_ids(const void* ids, int32_t ids_i64, int64_t num_rows, int64_t n_seg,
                                int64_t* seg_ptr, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6 / K7  row gather / scatter.  Replace paddle.gather (pgl/utils/op.py:45 read_rows / RowReader,
 * pgl/message.py:157 edge_expand, pgl/graph.py:822) and paddle.scatter(overwrite=True)
 * (pgl/graph.py:828-830 recv epilogue).  index int32 (index_i64 = 0) or int64 (= 1).
 *   gather : out[i, :] = x[index[i], :]          out [n_index, d]
 *   scatter: out[index[i], :] = x[i, :]          (indices unique; out pre-initialised by caller)
 * elem_bytes = size of one element (1,2,4,8): these are pure byte moves.
 * ---------------------------------------------------------------------------------------------- */
int32_t pglamd_gather_rows(const void* x, int64_t d, int32_t elem_bytes, const void* index,
                           int32_t index_i64, int64_t n_index, void* out, void* stream);
int32_t pglamd_scatter_rows(const void* x, int64_t d, int32_t elem_bytes, const void* index,
                            int32_t index_i64, int64_t n_index, void* out, void* stream);

/* K6w  row gather with a dtype change -- the wire pack / unpack of the halo exchange (16-bit wire for fp32 features):
 *   out[i, :] = cast(x[index[i], :]),  index int32 [n_index] or NULL (identity: a row-wise conversion of n_index rows).
 *   (x_dtype, out_dtype): F32 -> F16 | BF16 | F32, F16 | BF16 -> F32, F16 -> F16, BF16 -> BF16 (a plain pack of a column block).
 *   Stands where the reference would paddle.gather + cast.
 *   ldx: row stride of x in elements (0 = d): packs a column block of a wider matrix; out is dense [n_index, d]. */
int32_t pglamd_gather_rows_cast(const void* x, int32_t x_dtype, int64_t d, int64_t ldx, const int32_t* index,
                                int64_t n_index, void* out, int32_t out_dtype, void* stream);

/* K9  degree_norm.  Replaces cast/clip/pow in GF.degree_norm (pgl/nn/functional/graph_op.py:46-55):
 *     out[i] = max((float)degree[i], 1) ** -0.5      out F32 (out_f64 = 0) or F64 (= 1)           */
int32_t pglamd_degree_norm(const int64_t* degree, int64_t n, void* out, int32_t out_f64,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Neighbour sampling + relabel on the GPU ("next" row f3; the step before the hot path in the
 * GraphSAGE mini-batch loop).  Replaces paddle.geometric.sample_neighbors / reindex_graph as used
 * by pgl.sampling.NeighborSampler (pgl/sampling/sage.py:130-155) and the CPU path
 * Graph.sample_predecessor -> graph_kernel.sample_subset(_with_eid) (pgl/graph_kernel.pyx:266-339).
 *   count:  count[i] = min(degree(nodes[i]), k)   (k < 0: all neighbours)
 *   fill :  neighbours of nodes[i] written at offsets[i] (exclusive scan of count, made by the
 *           caller): all of them if degree <= k, else k uniformly WITHOUT replacement (Floyd),
 *           randomness = hash(seed, node, draw).  indptr/col/eid = the dst-sorted CSR; k <= 64.
 *   reindex: out_nodes = nodes, then every new neighbour id in order of first appearance;
 *           reindex_src[j] = new id of neighbors[j]; *num_out = len(out_nodes).  Deterministic.
 * ---------------------------------------------------------------------------------------------- */
int32_t pglamd_sample_neighbors_count(const int64_t* indptr, const int64_t* nodes, int64_t n,
                                      int64_t k, int64_t* count, void* stream);
int32_t pglamd_sample_neighbors_fill(const int64_t* indptr, const int32_t* col, const int32_t* eid,
                                     const int64_t* nodes, int64_t n, int64_t k, uint64_t seed,
                                     const int64_t* offsets, int64_t* out_neighbors,
                                     int64_t* out_eids, void* stream);
size_t pglamd_reindex_workspace_bytes(int64_t num_nodes, int64_t num_neighbors);
int32_t pglamd_reindex(const int64_t* nodes, int64_t num_nodes, const int64_t* neighbors,
                       int64_t num_neighbors, int64_t* reindex_src, int64_t* out_nodes,
                       int64_t* num_out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layer epilogue (row f1, "SpMM -> GEMM epilogue fusion"): the element passes after the dense X.W of
 * GraphSageConv (self + neigh + biases -> act -> F.normalize, pgl/nn/conv.py:99-115) and GCNConv (+ bias -> activation,
 * pgl/nn/conv.py:250-254) as ONE row kernel forward and one backward.  float32, rows contiguous.
 *   y[r,:] = normalize_L2( act( z[r,:] + bias ) )     act 0 none / 1 relu; normalize 0/1 (x / max(||x||, eps));
 *            y may alias z; inv_norm[n_rows] (written when normalize) is what the backward needs besides y.
 *   backward: dz = act'( normalize ? (dy - y <dy,y>) * inv_norm : dy ); col_partials[pglamd_row_epilogue_partials(n_rows), d]
 *            (optional) receives per-wave column sums of dz -- summed over rows they are the bias gradient.
 * ---------------------------------------------------------------------------------------------- */
int64_t pglamd_row_epilogue_partials(int64_t n_rows);
int32_t pglamd_row_epilogue(const float* z, const float* bias, int64_t n_rows, int64_t d, int32_t act,
                            int32_t normalize, float eps, float* y, float* inv_norm, void* stream);
int32_t pglamd_row_epilogue_backward(const float* dy, const float* y, const float* inv_norm,
                                     int64_t n_rows, int64_t d, int32_t act, int32_t normalize,
                                     float* dz, float* col_partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU exchange step of the row-partitioned path.  Replaces the collective of the reference's
 * DistGPUGraph -- all_reduce_sum_with_grad of the whole [N, d] output after every aggregation
 * (pgl/graph.py:1517-1553 -> pgl/utils/op.py:90-122, c_allreduce_sum) -- with one all-to-all-v of halo rows:
 *   pglamd_comm_unique_id   rank 0 fills id_out[PGLAMD_COMM_ID_BYTES]; the caller hands it to every rank by any means
 *                           (the reference's launcher broadcasts endpoints the same way: paddle.distributed.init_parallel_env)
 *   pglamd_comm_init        one RCCL communicator per process + a library-owned side stream + two events
 *   pglamd_halo_exchange_start   send_buf = rows for peer 0, 1, ... (send_rows[q] rows of row_bytes each), recv_buf
 *                           likewise.  Everything already queued on compute_stream (the pack kernel) is ordered before
 *                           the transfers by an event; the call returns at once, the caller keeps launching the
 *                           local-source aggregation on compute_stream -- the overlap.
 *   pglamd_halo_exchange_wait    makes compute_stream wait (on the GPU, no host sync) until recv_buf is complete.
 * RCCL is opened with dlopen at the first call: PGLAMD_E_UNAVAILABLE without librccl.so, PGLAMD_E_RCCL on RCCL errors.
 * The plan (which rows go where) is pglamd_halo_plan_* below.  pgl_amd.distributed uses torch.distributed's RCCL
 * all_to_all_single by default and this transport with PGLAMD_TRANSPORT=abi; a caller without torch has only this one.
 * ---------------------------------------------------------------------------------------------- */
#define PGLAMD_COMM_ID_BYTES 128
int32_t pglamd_comm_unique_id(void* id_out);
int32_t pglamd_comm_init(int32_t rank, int32_t world, const void* unique_id, void** comm_out);
int32_t pglamd_comm_destroy(void* comm);
int32_t pglamd_halo_exchange_start(void* comm, const void* send_buf, const int64_t* send_rows,
                                   void* recv_buf, const int64_t* recv_rows, int64_t row_bytes,
                                   void* compute_stream);
int32_t pglamd_halo_exchange_wait(void* comm, void* compute_stream);
/* The exchange WITHOUT a send buffer (ABI 3): with the owner's rows ordered by the set of peers that pull them
 * (pgl_amd.distributed.HaloPlan(row_order="peers")) every peer's rows are a few contiguous ranges of the feature matrix `x` itself;
 * they are sent from where they lie (one ncclSend per range) into the matching ranges of recv_buf -- no pack launch, no send-buffer
 * traffic.  send_ptr / recv_ptr [world + 1] delimit each peer's ranges in send_first / send_cnt (rows of x) and recv_first /
 * recv_cnt (rows of recv_buf); range k of a pair has the same length on both ends.  Several exchanges (up to 8) may be in flight;
 * pglamd_halo_exchange_wait retires them in start order. */
int32_t pglamd_halo_exchange_start_ranges(void* comm, const void* x, const int64_t* send_ptr,
                                          const int64_t* send_first, const int64_t* send_cnt, void* recv_buf,
                                          const int64_t* recv_ptr, const int64_t* recv_first,
                                          const int64_t* recv_cnt, int64_t row_bytes, void* compute_stream);

/* Halo plan of one rank (HOST pointers; every rank derives its own share from the global edge list and the part vector,
 * so no negotiation is needed).  Relabelling and layout follow apps/GNNAutoScale/graph_partition.py:70-101 (owned ids
 * contiguous per part, ascending original id inside a part) and apps/GNNAutoScale/dataset.py:196-209 ([owned | halo]).
 *   pglamd_halo_plan_sizes  -> sizes[5] = n_own, n_local_source_edges, n_halo_source_edges, n_halo_rows, n_send_rows
 *   pglamd_halo_plan_fill   offsets[world+1] first relabelled id of every part; own_global[n_own] original id of local row i;
 *                           (loc_rows, loc_cols)[n_loc] local dst row / local src row of the edges with both ends owned;
 *                           (hal_rows, hal_cols)[n_hal] local dst row / position in the halo block of the other in-edges;
 *                           halo_global[n_halo] relabelled ids of the halo rows, ascending = grouped by owner;
 *                           send_idx[n_send] owned rows to pack, grouped by destination rank; halo_splits / pull_splits[world]
 *                           rows received from / sent to each rank (the recv_rows / send_rows of pglamd_halo_exchange_start);
 *                           in_degree / out_degree[n_own] GLOBAL degrees of the owned nodes (DistGPUGraph.indegree/outdegree,
 *                           pgl/graph.py:1524-1532); edge_global[n_loc + n_hal] original edge id of local edge k
 *                           (order: local-source edges, then halo-source edges) -- where edge features are sliced.
 *   Any output pointer except offsets / halo_splits / pull_splits may be NULL.  Bit-identical to pgl_amd.distributed.HaloPlan. */
int32_t pglamd_halo_plan_sizes(const int64_t* src, int64_t src_stride, const int64_t* dst,
                               int64_t dst_stride, int64_t num_edges, int64_t num_nodes,
                               const int64_t* part, int32_t rank, int32_t world, int64_t* sizes);
int32_t pglamd_halo_plan_fill(const int64_t* src, int64_t src_stride, const int64_t* dst,
                              int64_t dst_stride, int64_t num_edges, int64_t num_nodes,
                              const int64_t* part, int32_t rank, int32_t world, int64_t* offsets,
                              int64_t* own_global, int64_t* loc_rows, int64_t* loc_cols,
                              int64_t* hal_rows, int64_t* hal_cols, int64_t* halo_global,
                              int64_t* send_idx, int64_t* halo_splits, int64_t* pull_splits,
                              int64_t* in_degree, int64_t* out_degree, int64_t* edge_global);

/* ------------------------------------------------------------------------------------------------
 * Host-side (CPU) helpers.  Pointers here are HOST pointers.
 * pglamd_map_ids replaces graph_kernel.map_edges / map_nodes (pgl/graph_kernel.pyx:104-138):
 *     out[i] = value of key in[i] in the (keys -> vals) dictionary; missing key -> 0, like
 *     std::unordered_map::operator[] in the reference.
 * pglamd_partition_kway stands behind pgl.partition.metis_partition (pgl/partition.py:37-91 ->
 * graph_kernel.metis_partition, pgl/graph_kernel.pyx:434-472).  No METIS code is built into or opened by
 * this library (round 5: the round-3/4 opt-in bridge pglamd_partition_metis / libpglamd_metis.so was removed from
 * the product; the reference's METIS lives only in the test oracle, oracle/_ref).
 * It is the engine's own multilevel k-way partitioner.  Same inputs as METIS_PartGraphKway as the reference calls it (CSR xadj/adjncy
 * int64, optional positive int64 vertex / edge weights), same output (part[N] int64 in
 * [0, nparts)); its ids are NOT METIS's -- parity there is on balance and edge cut.
 * ---------------------------------------------------------------------------------------------- */
/* Host twin of pglamd_csr_build for numpy-mode graphs (Graph.indegree()/sorted_edges() before
 * Graph.tensor(), as examples/gcn/train.py:83 does): same outputs, same order, HOST pointers.
 * Replaces graph_kernel.build_index (pgl/graph_kernel.pyx:59-88) on the CPU side. */
int32_t pglamd_build_index_host(const int64_t* u, int64_t u_stride, const int64_t* v,
                                int64_t v_stride, int64_t num_edges, int64_t num_nodes,
                                int64_t* degree, int64_t* sorted_v, int64_t* sorted_u,
                                int64_t* sorted_eid, int64_t* indptr);
int32_t pglamd_map_ids(const int64_t* keys, const int64_t* vals, int64_t n_keys, const int64_t* in,
                       int64_t n_in, int64_t* out);
int32_t pglamd_partition_kway(int64_t num_nodes, const int64_t* xadj, const int64_t* adjncy,
                              const int64_t* vwgt, const int64_t* adjwgt, int64_t nparts,
                              uint64_t seed, int64_t* part, int64_t* edgecut);
/* The same partitioner with what a row-partitioned aggregation needs: a SECOND balance constraint (vwgt2, e.g. 1 per row next to
 * vwgt = in-degree + 1: a rank's time follows its edges, its row-wise kernels and its memory follow its rows), explicit imbalance
 * bounds ub / ub2 (<= 1: 1.03) and a thread count (0: $PGLAMD_THREADS, else min(cores, 16)).  Parallel and deterministic: the
 * parts depend on (graph, weights, nparts, seed) only, not on the thread count.  pglamd_partition_edges takes the DIRECTED edge
 * list as it is (strided int64 src / dst) and builds the symmetrised adjacency itself, in parallel. */
int32_t pglamd_partition_kway2(int64_t num_nodes, const int64_t* xadj, const int64_t* adjncy,
                               const int64_t* vwgt, const int64_t* vwgt2, const int64_t* adjwgt,
                               int64_t nparts, double ub, double ub2, uint64_t seed, int32_t threads,
                               int64_t* part, int64_t* edgecut);
int32_t pglamd_partition_edges(const int64_t* src, int64_t src_stride, const int64_t* dst,
                               int64_t dst_stride, int64_t num_edges, int64_t num_nodes,
                               const int64_t* vwgt, const int64_t* vwgt2, int64_t nparts, double ub,
                               double ub2, uint64_t seed, int32_t threads, int64_t* part,
                               int64_t* edgecut);

#ifdef __cplusplus
}
#endif
#endif /* PGL_AMD_H_ */
