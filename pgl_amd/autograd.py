"""Differentiable wrappers (torch.autograd.Function) around pgl_amd.ops.

The C ABI is forward-only and stateless; gradients are expressed with the SAME
kernels on the transposed index (adj_src_index): d/dx of a sum over in-edges is a sum over
out-edges.  Fused backward kernels (SDDMM for d/d(edge feature), softmax-backward) are the
"next" row f1; where they are missing the backward composes existing kernels and says so.
"""
import torch

from . import ops
from . import edge_tensor as _et


def _unbroadcast(g, shape):
    """Sum `g` ([rows, *out_tail]) back to [rows, *tail] after numpy-style broadcasting."""
    tail = tuple(shape[1:])
    gt = tuple(g.shape[1:])
    if gt == tail:
        return g
    lead = len(gt) - len(tail)
    if lead > 0:
        g = g.sum(dim=tuple(range(1, 1 + lead)))
    dims = tuple(i + 1 for i, s in enumerate(tail) if s == 1 and g.shape[i + 1] != 1)
    if dims:
        g = g.sum(dim=dims, keepdim=True)
    return g.reshape((g.shape[0],) + tail)


class _Aggregate(torch.autograd.Function):
    """send_u_recv / send_ue_recv.  csr = dst-keyed CSR; csr_t() lazily returns the src-keyed one."""

    @staticmethod
    def forward(ctx, x, y, csr, csr_t, rop, mop, out_size, src32, dst32, src_scale=None, dst_scale=None):
        out = ops.aggregate(x, csr, rop, out_size, y, mop, src_scale, dst_scale)
        ctx.csr, ctx.csr_t, ctx.rop, ctx.mop = csr, csr_t, rop, mop
        ctx.scales = (src_scale, dst_scale)
        ctx.src32, ctx.dst32 = src32, dst32
        ctx.x_shape = tuple(x.shape)
        ctx.y_shape = None if y is None else tuple(y.shape)
        need_out = rop in ("max", "min")
        ctx.save_for_backward(x, y if y is not None else x.new_zeros(0), out if need_out else x.new_zeros(0))
        return out

    @staticmethod
    def backward(ctx, grad):
        x, y, out = ctx.saved_tensors
        has_y = ctx.y_shape is not None
        y = y if has_y else None
        grad = grad.contiguous()
        gx = gy = None
        csr_t = ctx.csr_t()
        n_x = ctx.x_shape[0]
        if ctx.scales[0] is not None or ctx.scales[1] is not None:
            # out = ds * A (ss * x)  =>  dx = ss * A^T (ds * g): the two fused scales swap roles
            g = ops.aggregate(grad, csr_t, "sum", n_x, src_scale=ctx.scales[1], dst_scale=ctx.scales[0])
            return (g,) + (None,) * 10
        if ctx.rop in ("sum", "mean"):
            scale = None
            if ctx.rop == "mean":
                # d out[v] / d msg = 1 / indeg(v): rides along as the per-source scale of the transposed sum
                scale = (1.0 / ctx.csr.degree.clamp(min=1).to(torch.float32))
                if grad.shape[0] > scale.shape[0]:
                    scale = torch.cat([scale, scale.new_ones(grad.shape[0] - scale.shape[0])])
                if grad.dtype != torch.float32:       # kernel scales are fp32-only: pre-scale instead
                    grad = grad * scale.to(grad.dtype).reshape((-1,) + (1,) * (grad.dim() - 1))
                    scale = None
            if ctx.needs_input_grad[0]:
                if not has_y or ctx.mop in ("add", "sub"):
                    g = ops.aggregate(grad, csr_t, "sum", n_x, src_scale=scale)
                elif ctx.mop == "mul":
                    g = ops.aggregate(grad, csr_t, "sum", n_x, y, "mul", src_scale=scale)
                else:
                    g = ops.aggregate(grad, csr_t, "sum", n_x, y, "div", src_scale=scale)
                gx = _unbroadcast(g, ctx.x_shape)
            if (has_y and ctx.needs_input_grad[1] and ctx.mop == "mul" and scale is None and grad.dtype == torch.float32
                    and grad.dim() == 3 and len(ctx.y_shape) == 3 and ctx.y_shape[2] == 1 and ctx.y_shape[1] == grad.shape[1]
                    and tuple(ctx.x_shape[1:]) == tuple(grad.shape[1:]) and ops.sddmm_supported(grad.shape[1], grad.shape[2])):
                # d/de of sum_e x[src] * e with e [E,H,1]: one SDDMM pass, no [E,H,D] gather is materialised
                gy = ops.sddmm(x, grad, ctx.csr).reshape(ctx.y_shape)
            elif (has_y and ctx.needs_input_grad[1] and grad.shape[0] >= ctx.csr.num_nodes
                  and ops.edge_operand_grad_supported(grad, x if tuple(x.shape[1:]) == tuple(grad.shape[1:]) else grad, ctx.y_shape)
                  and (ctx.mop in ("add", "sub") or tuple(x.shape[1:]) == tuple(grad.shape[1:]))):
                # trailing-dim broadcast operands ([E], [E,1], [E,d], [E,H,D]): one pass over the edges, the [E, d] products
                # live in registers only (round 3; the composition below materialises three [E, d] tensors)
                gy = ops.edge_operand_grad(grad, x, y if ctx.mop == "div" else None, ctx.csr, ctx.mop, ctx.y_shape, scale)
            elif has_y and ctx.needs_input_grad[1]:
                # general broadcast shapes: composed from row gathers (materialises [E, out_tail])
                gd = ops.gather_rows(grad, ctx.dst32)
                if scale is not None:
                    gd = gd * ops.gather_rows(scale.reshape(-1, 1), ctx.dst32).reshape((-1,) + (1,) * (gd.dim() - 1))
                if ctx.mop == "add":
                    gy = gd
                elif ctx.mop == "sub":
                    gy = -gd
                else:
                    xs = ops.gather_rows(x, ctx.src32)
                    gy = gd * xs if ctx.mop == "mul" else -gd * xs / (y * y)
                gy = _unbroadcast(gy, ctx.y_shape)
        elif not has_y and ops.winner_grad_supported(x, out) and grad.shape[0] == n_x == out.shape[0]:
            # max / min without an edge operand (GraphSage's pooling): the winner mask is evaluated inside ONE walk of the
            # src-sorted stream -- no [E, d] tensor (round 3)
            if ctx.needs_input_grad[0]:
                gx = ops.winner_grad(grad, out, x, csr_t)
        else:
            # max / min with an edge operand: gradient flows to every message equal to the winner (Paddle's rule);
            # composed from gathers
            xs = ops.gather_rows(x, ctx.src32)
            msg = xs if not has_y else {"add": xs + y, "sub": xs - y, "mul": xs * y, "div": xs / y}[ctx.mop]
            hit = (msg == ops.gather_rows(out, ctx.dst32)).to(grad.dtype)
            gm = ops.gather_rows(grad, ctx.dst32) * hit
            if ctx.needs_input_grad[0]:
                gxe = gm if not has_y or ctx.mop in ("add", "sub") else (gm * y if ctx.mop == "mul" else gm / y)
                g = ops.aggregate(gxe.contiguous(), _edge_csr(csr_t), "sum", n_x)
                gx = _unbroadcast(g, ctx.x_shape)
            if has_y and ctx.needs_input_grad[1]:
                gy = {"add": gm, "sub": -gm, "mul": gm * xs, "div": -gm * xs / (y * y)}[ctx.mop]
                gy = _unbroadcast(gy, ctx.y_shape)
        return gx, gy, None, None, None, None, None, None, None, None, None


class _EdgeCSR(object):
    """View of a CSR whose 'source rows' are EDGE rows: col = original edge id."""

    def __init__(self, c):
        self.row32, self.col32, self.eid32, self.indptr = c.row32, c.eid32, c.eid32, c.indptr
        self.num_edges, self.num_nodes, self.degree = c.num_edges, c.num_nodes, c.degree


def _edge_csr(c):
    return _EdgeCSR(c)


def aggregate(x, csr, csr_t, reduce_op="sum", out_size=None, y=None, message_op="add", src32=None, dst32=None,
              src_scale=None, dst_scale=None):
    """src_scale / dst_scale ([N] fp32, no gradient): fused row scalings, only with y=None and sum."""
    y = _et.materialize(y)                 # (an EdgeTensor reaching this level is read in original edge order)
    if torch.is_grad_enabled() and (x.requires_grad or (y is not None and y.requires_grad)):
        return _Aggregate.apply(x, y, csr, csr_t, reduce_op, message_op, out_size, src32, dst32, src_scale, dst_scale)
    return ops.aggregate(x, csr, reduce_op, out_size, y, message_op, src_scale, dst_scale)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, index, index_csr):
        ctx.index, ctx.index_csr, ctx.n = index, index_csr, x.shape[0]
        return ops.gather_rows(x, index)

    @staticmethod
    def backward(ctx, grad):
        csr = ctx.index_csr() if callable(ctx.index_csr) else ctx.index_csr
        if csr is None:     # arbitrary index: key it on the fly
            iota = torch.arange(ctx.index.shape[0], device=grad.device, dtype=torch.int64)
            csr = ops.csr_build(ctx.index.to(torch.int64), iota, ctx.n, check_range=False)
        return ops.aggregate(grad.contiguous(), _edge_csr(csr), "sum", ctx.n), None, None


def gather_rows(x, index, index_csr=None):
    """Differentiable paddle.gather(x, index, axis=0).  index_csr: CSR keyed by `index` (the graph
    passes adj_src/adj_dst so the backward is one aggregation, no sort)."""
    x = _et.materialize(x)
    if torch.is_grad_enabled() and x.requires_grad:
        return _GatherRows.apply(x, index, index_csr)
    return ops.gather_rows(x, index)


class _SendUV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, src32, dst32, mop, csr_dst, csr_src):
        ctx.mop, ctx.src32, ctx.dst32, ctx.csr_dst, ctx.csr_src = mop, src32, dst32, csr_dst, csr_src
        ctx.save_for_backward(x, y)
        return ops.send_uv(x, y, src32, dst32, mop)

    @staticmethod
    def backward(ctx, grad):
        x, y = ctx.saved_tensors
        grad = grad.contiguous()
        gx = gy = None
        if ctx.mop in ("add", "sub"):
            ge_x, ge_y = grad, (grad if ctx.mop == "add" else -grad)
        else:
            xs, yd = ops.gather_rows(x, ctx.src32), ops.gather_rows(y, ctx.dst32)
            xs = xs.reshape((xs.shape[0],) + (1,) * (grad.dim() - xs.dim()) + tuple(xs.shape[1:]))
            yd = yd.reshape((yd.shape[0],) + (1,) * (grad.dim() - yd.dim()) + tuple(yd.shape[1:]))
            if ctx.mop == "mul":
                ge_x, ge_y = grad * yd, grad * xs
            else:
                ge_x, ge_y = grad / yd, -grad * xs / (yd * yd)
        if ctx.needs_input_grad[0]:
            g = ops.aggregate(ge_x.contiguous(), _edge_csr(ctx.csr_src()), "sum", x.shape[0])
            gx = _unbroadcast(g, tuple(x.shape))
        if ctx.needs_input_grad[1]:
            g = ops.aggregate(ge_y.contiguous(), _edge_csr(ctx.csr_dst()), "sum", y.shape[0])
            gy = _unbroadcast(g, tuple(y.shape))
        return gx, gy, None, None, None, None, None


def send_uv(x, y, src32, dst32, mop, csr_dst, csr_src):
    if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad):
        return _SendUV.apply(x, y, src32, dst32, mop, csr_dst, csr_src)
    return ops.send_uv(x, y, src32, dst32, mop)


class _SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, ids, pool, num_segments):
        out = ops.segment_reduce(data, ids, pool, num_segments)
        ctx.pool, ctx.ids = pool, ids
        ctx.save_for_backward(data, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        data, out = ctx.saved_tensors
        grad = grad.contiguous()
        ge = ops.gather_rows(grad, ctx.ids)
        if ctx.pool == "mean":
            seg_ptr = ops.seg_ptr_from_ids(ctx.ids, out.shape[0])
            cnt = (seg_ptr[1:] - seg_ptr[:-1]).clamp(min=1).to(grad.dtype)
            ge = ge / ops.gather_rows(cnt.reshape(-1, 1), ctx.ids).reshape((-1,) + (1,) * (ge.dim() - 1))
        elif ctx.pool in ("max", "min"):
            ge = ge * (data == ops.gather_rows(out, ctx.ids)).to(grad.dtype)
        return ge, None, None, None


def segment_reduce(data, ids, pool="sum", num_segments=None):
    data = _et.materialize(data)
    if not isinstance(data, torch.Tensor) or not isinstance(ids, torch.Tensor):
        raise TypeError("pgl.math.segment_%s takes device tensors (data: %s, segment_ids: %s); there is no host path"
                        % (pool, type(data).__name__, type(ids).__name__))
    if torch.is_grad_enabled() and data.requires_grad:
        return _SegmentReduce.apply(data, ids, pool, num_segments)
    return ops.segment_reduce(data, ids, pool, num_segments)


class _SegView2CSR(object):
    """Adapter: lets ops.aggregate run a segment reduction described by an ops.SegView."""

    def __init__(self, v, n_elem):
        self.row32, self.col32, self.eid32, self.indptr = v.row32, v.perm32, v.perm32, v.seg_ptr
        self.num_edges, self.num_nodes = n_elem, int(v.seg_ptr.shape[0]) - 1


class _SegmentSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, view):
        out = ops.segment_softmax(data, view)
        ctx.view = view
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad):
        # dL/dx = p * (g - sum_seg(p * g)): the segment sum runs through the flat aggregation kernel
        (p,) = ctx.saved_tensors
        v = ctx.view
        pg = (p * grad).contiguous()
        n_seg = int(v.seg_ptr.shape[0]) - 1
        s = ops.aggregate(pg, _SegView2CSR(v, int(p.shape[0])), "sum", n_seg)
        return pg - p * ops.gather_rows(s, v.elem_seg32), None


def segment_softmax(data, view):
    data = _et.materialize(data)
    if torch.is_grad_enabled() and data.requires_grad:
        return _SegmentSoftmax.apply(data, view)
    return ops.segment_softmax(data, view)


class _GatAttention(torch.autograd.Function):
    """Fused GAT attention aggregation (forward: one pass, online softmax; backward: alpha recomputed
    from the saved per-row statistics).  Attention dropout is regenerated from (seed, edge id, head)."""

    @staticmethod
    def forward(ctx, feature, attn_src, attn_dst, csr_dst, csr_src_fn, slope, drop_p, seed):
        out, mx, sm, out_pos, s_pos = ops.gat_aggregate(feature, attn_src, attn_dst, csr_dst, slope, None, True, drop_p, seed)
        ctx.csr_dst, ctx.csr_src_fn, ctx.slope, ctx.drop_p, ctx.seed = csr_dst, csr_src_fn, slope, drop_p, seed
        ctx.has_pos = out_pos is not None
        ctx.save_for_backward(feature, attn_src, attn_dst, out, mx, sm, *((out_pos, s_pos) if ctx.has_pos else ()))
        return out

    @staticmethod
    def backward(ctx, grad):
        feature, a_s, a_d, out, mx, sm = ctx.saved_tensors[:6]
        out_pos, s_pos = ctx.saved_tensors[6:] if ctx.has_pos else (None, None)
        gf, gs, gd = ops.gat_backward(grad, feature, out, a_s, a_d, mx, sm, ctx.csr_dst, ctx.csr_src_fn(), ctx.slope,
                                      ctx.drop_p, ctx.seed, out_pos, s_pos)
        return gf, gs, gd, None, None, None, None, None


def gat_attention(feature, attn_src, attn_dst, csr_dst, csr_src_fn, slope=0.2, drop_p=0.0, seed=0):
    if torch.is_grad_enabled() and (feature.requires_grad or attn_src.requires_grad or attn_dst.requires_grad):
        return _GatAttention.apply(feature, attn_src, attn_dst, csr_dst, csr_src_fn, slope, drop_p, seed)
    return ops.gat_aggregate(feature, attn_src, attn_dst, csr_dst, slope, None, False, drop_p, seed)


class _GatAttentionProj(torch.autograd.Function):
    """_GatAttention with the two score projections inside the node: a_src | a_dst = feature2d @ proj (proj [H*D, 2H]: GATConv's
    block-diagonal form of weight_src / weight_dst, pgl/nn/conv.py:325-330).  As separate nodes `feature` receives two gradients -- from
    the attention kernels and from the projection GEMM -- which autograd adds with an element pass over [N, H*D] (0.27 ms at N = 2^20,
    128 columns); here the projection's share is ACCUMULATED into the kernels' output by the GEMM itself (addmm_, beta = 1)."""

    @staticmethod
    def forward(ctx, feature, proj, csr_dst, csr_src_fn, slope, drop_p, seed):
        n, h = int(feature.shape[0]), int(feature.shape[1])
        # (the same GEMM call as the three-node form makes -- linear(x, proj^T) -- so the scores are the same BITS: leaky_relu has a kink at
        #  0 and an edge whose score lands within rounding of it would otherwise take a different slope in the two forms)
        att = torch.nn.functional.linear(feature.reshape(n, -1), proj.t().contiguous())
        a_s, a_d = att[:, :h].contiguous(), att[:, h:].contiguous()
        out, mx, sm, out_pos, s_pos = ops.gat_aggregate(feature, a_s, a_d, csr_dst, slope, None, True, drop_p, seed)
        ctx.csr_dst, ctx.csr_src_fn, ctx.slope, ctx.drop_p, ctx.seed = csr_dst, csr_src_fn, slope, drop_p, seed
        ctx.has_pos = out_pos is not None
        ctx.save_for_backward(feature, proj, a_s, a_d, out, mx, sm, *((out_pos, s_pos) if ctx.has_pos else ()))
        return out

    @staticmethod
    def backward(ctx, grad):
        feature, proj, a_s, a_d, out, mx, sm = ctx.saved_tensors[:7]
        out_pos, s_pos = ctx.saved_tensors[7:] if ctx.has_pos else (None, None)
        gf, gs, gd = ops.gat_backward(grad, feature, out, a_s, a_d, mx, sm, ctx.csr_dst, ctx.csr_src_fn(), ctx.slope,
                                      ctx.drop_p, ctx.seed, out_pos, s_pos)
        n = int(feature.shape[0])
        datt = torch.cat([gs, gd], dim=1)
        gproj = _tall_wgrad(datt, feature.reshape(n, -1)).t() if ctx.needs_input_grad[1] else None
        if ctx.needs_input_grad[0]:
            gf.reshape(n, -1).addmm_(datt, proj.t())                 # d feature += d att @ proj^T, inside the GEMM
        else:
            gf = None
        return gf, gproj, None, None, None, None, None


def gat_attention_proj(feature, proj, csr_dst, csr_src_fn, slope=0.2, drop_p=0.0, seed=0):
    return _GatAttentionProj.apply(feature, proj, csr_dst, csr_src_fn, slope, drop_p, seed)


class _SDDMM(torch.autograd.Function):
    """out[e, h] = <x[src_e, h, :], y[dst_e, h, :]> in original edge order.  Backward = two edge-weighted aggregations
    (the same flat kernel with an [E,H,1] edge operand): d x[u] = sum_{e: src=u} g_e y[dst_e] over the src-keyed CSR,
    d y[v] = sum_{e: dst=v} g_e x[src_e] over the dst-keyed one.  No [E,H,D] tensor in either direction."""

    @staticmethod
    def forward(ctx, x, y, csr_dst, csr_src_fn):
        ctx.csr_dst, ctx.csr_src_fn = csr_dst, csr_src_fn
        ctx.save_for_backward(x, y)
        return ops.sddmm(x, y, csr_dst)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.contiguous().reshape(g.shape[0], g.shape[1], 1)
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = ops.aggregate(y, ctx.csr_src_fn(), "sum", int(x.shape[0]), g, "mul")
        if ctx.needs_input_grad[1]:
            gy = ops.aggregate(x, ctx.csr_dst, "sum", int(y.shape[0]), g, "mul")
        return gx, gy, None, None


def sddmm(x, y, csr_dst, csr_src_fn):
    if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad):
        return _SDDMM.apply(x, y, csr_dst, csr_src_fn)
    return ops.sddmm(x, y, csr_dst)


class _PropagateStep(torch.autograd.Function):
    """out = c * res + dst_scale (.) A x  in one aggregation launch: `out` starts as c * res and the kernel adds the
    scaled row sums into it (accumulate mode).  The k-hop layers (APPNP, SGC, SSGC, GPR) iterate on g = norm (.) h, for
    which one hop of  norm (.) A (norm (.) h)  is exactly this with dst_scale = norm^2 -- no per-hop scaling passes."""

    @staticmethod
    def forward(ctx, x, res, c, dst_scale, csr, csr_t_fn):
        ctx.csr_t_fn, ctx.c, ctx.has_res = csr_t_fn, c, res is not None
        ctx.save_for_backward(dst_scale)
        ctx.n_x = int(x.shape[0])
        if res is None:
            return ops.aggregate(x, csr, "sum", None, dst_scale=dst_scale)
        out = res * c
        return ops.aggregate(x, csr, "sum", None, dst_scale=dst_scale, out=out, accumulate=True)

    @staticmethod
    def backward(ctx, grad):
        (dst_scale,) = ctx.saved_tensors
        grad = grad.contiguous()
        gx = ops.aggregate(grad, ctx.csr_t_fn(), "sum", ctx.n_x, src_scale=dst_scale) if ctx.needs_input_grad[0] else None
        gres = grad * ctx.c if (ctx.has_res and ctx.needs_input_grad[1]) else None
        return gx, gres, None, None, None, None


def propagate_step(x, dst_scale, csr, csr_t_fn, res=None, c=0.0):
    if torch.is_grad_enabled() and (x.requires_grad or (res is not None and res.requires_grad)):
        return _PropagateStep.apply(x, res, c, dst_scale, csr, csr_t_fn)
    if res is None:
        return ops.aggregate(x, csr, "sum", None, dst_scale=dst_scale)
    return ops.aggregate(x, csr, "sum", None, dst_scale=dst_scale, out=res * c, accumulate=True)


class _AddScore(torch.autograd.Function):
    """s[e, h] = sum_d w[h, d] * leaky(x[src_e] + y[dst_e]): one pass forward; backward is one walk per orientation of the
    edge list (d y and d w over the dst-keyed index, d x over the src-keyed one)."""

    @staticmethod
    def forward(ctx, x, y, w, csr_dst, csr_src_fn, slope):
        ctx.csr_dst, ctx.csr_src_fn, ctx.slope = csr_dst, csr_src_fn, slope
        ctx.save_for_backward(x, y, w)
        return ops.add_score(x, y, w, csr_dst, slope)

    @staticmethod
    def backward(ctx, g):
        x, y, w = ctx.saved_tensors
        g = g.contiguous()
        gx = gy = gw = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gy, gw = ops.add_score_backward(x, y, w, g, ctx.csr_dst, int(y.shape[0]), ctx.slope, want_w=ctx.needs_input_grad[2])
            if gw is not None:
                gw = gw.reshape(w.shape)
        if ctx.needs_input_grad[0]:
            gx, _ = ops.add_score_backward(y, x, w, g, ctx.csr_src_fn(), int(x.shape[0]), ctx.slope)
        return gx, gy, gw, None, None, None


def add_score(x, y, w, csr_dst, csr_src_fn, slope=0.2):
    if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad or w.requires_grad):
        return _AddScore.apply(x, y, w, csr_dst, csr_src_fn, slope)
    return ops.add_score(x, y, w, csr_dst, slope)


class _ScatterRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n_rows, index, x):
        out = torch.zeros((n_rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        ctx.index = index
        return ops.scatter_rows(out, index, x)

    @staticmethod
    def backward(ctx, grad):
        return None, None, ops.gather_rows(grad.contiguous(), ctx.index)


def scatter_into_zeros(n_rows, index, x):
    """zeros([n_rows, ...]) with rows `index` (unique) overwritten by x (pgl/graph.py:828-830)."""
    x = _et.materialize(x)
    if torch.is_grad_enabled() and x.requires_grad:
        return _ScatterRows.apply(n_rows, index, x)
    out = torch.zeros((n_rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    return ops.scatter_rows(out, index, x)


class _RowEpilogue(torch.autograd.Function):
    """y = normalize_L2(act(z + bias)): one kernel forward, one backward (which also yields the bias gradient)."""

    @staticmethod
    def forward(ctx, z, bias, act, normalize):
        y, inv = ops.row_epilogue(z, bias, act, normalize)
        ctx.act, ctx.normalize, ctx.has_bias = act, normalize, bias is not None
        ctx.save_for_backward(y, inv if inv is not None else y.new_zeros(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        dz, db = ops.row_epilogue_backward(dy, y, inv if ctx.normalize else None, ctx.act, ctx.normalize, want_b)
        return dz, db, None, None


def row_epilogue(z, bias=None, act=None, normalize=False):
    if torch.is_grad_enabled() and (z.requires_grad or (bias is not None and bias.requires_grad)):
        return _RowEpilogue.apply(z, bias, act, normalize)
    return ops.row_epilogue(z, bias, act, normalize)[0]


def column_sum(g):
    """g.sum(0) for [N, C] with N in the millions (bias gradients): two stages, [S, N / S, C] -> [S, C] -> [C].  The stock one-stage
    reduction picks 64 workgroups for a narrow odd C and takes 6.9 ms at N = 2^20, C = 41 (1.5 % of what the pass should cost);
    in two stages every stage has thousands of independent outputs."""
    n = g.shape[0]
    if g.dim() != 2 or n < 65536:
        return g.sum(0)
    g = g.contiguous()
    split = 1024
    m = n // split * split
    out = g[:m].view(split, m // split, -1).sum(1).sum(0)
    return out if m == n else out + g[m:].sum(0)


def _tall_wgrad(g, x):
    """g^T x for [N, out] / [N, in] with N in the millions: the reduction over N split into 256 batched slabs (bmm -> sum); the
    stock GEMM runs the whole reduction serially per output tile (1.9 ms vs 0.33 ms at N = 2^20, 128 x 128)."""
    n = x.shape[0]
    if n < 65536:
        return g.t() @ x
    split = 256
    m = n // split * split
    gw = torch.bmm(g[:m].view(split, m // split, -1).transpose(1, 2), x[:m].view(split, m // split, -1)).sum(0)
    if m < n:
        gw = gw + g[m:].t() @ x[m:]
    return gw


class _AggregateDense(torch.autograd.Function):
    """out = act( (dst_scale * A (src_scale * x)) @ W^T + b ) in one kernel (ops.aggregate_dense).  `weight` is nn.Linear's
    [d_out, d_in]; the scales carry no gradient.  Backward (fp32, same kernels): dZ = d out masked by the activation; d b = column
    sums of dZ; d W = dZ^T agg (the kept aggregate, split-reduction GEMM); d x = src_scale * A^T (dst_scale * (dZ W))
    = src_scale * ((A^T (dst_scale * dZ)) W) -- row scaling commutes with the right-multiplication -- i.e. the SAME fused kernel on the
    transposed index with W as the layer and src_scale as ITS per-row scale (no element pass of its own), when the width allows,
    else aggregate-then-GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, csr, csr_t, act, src_scale, dst_scale, reduce_op):
        need_w = weight.requires_grad or (bias is not None and bias.requires_grad)
        out, agg = ops.aggregate_dense(x, csr, weight.t(), bias, act, reduce_op, dst_scale, keep_agg=need_w, src_scale=src_scale)
        ctx.csr_t, ctx.act, ctx.reduce_op, ctx.csr = csr_t, act, reduce_op, csr
        ctx.has_bias = bias is not None
        ctx.n_x = int(x.shape[0])
        none = x.new_zeros(0)
        ctx.save_for_backward(weight, out if act == "relu" else none, agg if agg is not None else none,
                              dst_scale if dst_scale is not None else none, src_scale if src_scale is not None else none)
        return out

    @staticmethod
    def backward(ctx, g):
        weight, out, agg, ds, ss = ctx.saved_tensors
        g = g.contiguous()
        gw = gb = gx = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.act == "relu" and ops.row_epilogue_supported(g, int(g.shape[1])):
            # relu mask and the bias gradient's column sums in ONE pass over [N, d] (compare + multiply + sum as three torch
            # kernels: 0.45 ms at C2)
            dz, gb = ops.row_epilogue_backward(g, out, None, "relu", False, want_bias=want_b)
        else:
            dz = g * (out > 0).to(g.dtype) if ctx.act == "relu" else g
            if want_b:
                gb = column_sum(dz)
        if ctx.needs_input_grad[1]:
            gw = _tall_wgrad(dz, agg)
        if ctx.needs_input_grad[0]:
            scale = ds if ds.numel() else None
            if ctx.reduce_op == "mean":
                inv = 1.0 / (ctx.csr.indptr[1:] - ctx.csr.indptr[:-1]).clamp(min=1).to(torch.float32)
                scale = inv if scale is None else scale * inv
            csr_t = ctx.csr_t()
            back = ss if ss.numel() else None
            # (the destination scale of the forward is the per-SOURCE scale of the transposed walk: it rides along the transposed
            #  stream as one value per edge position -- no element pass over d Z)
            if ops.aggregate_dense_supported(dz, weight.shape[1]):
                gx = ops.aggregate_dense(dz, csr_t, weight, None, None, "sum", back, out_size=ctx.n_x, src_scale=scale)[0]
            else:
                gx = ops.aggregate(dz, csr_t, "sum", ctx.n_x, src_scale=scale, dst_scale=back) @ weight
        return gx, gw, gb, None, None, None, None, None, None


def aggregate_dense(x, weight, bias, csr, csr_t, act=None, dst_scale=None, reduce_op="sum", src_scale=None):
    """x [N, d_in] fp32, weight [d_out, d_in] (nn.Linear layout), bias [d_out] or None; src_scale / dst_scale one value per node."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _AggregateDense.apply(x, weight, bias, csr, csr_t, act, src_scale, dst_scale, reduce_op)
    return ops.aggregate_dense(x, csr, weight.t(), bias, act, reduce_op, dst_scale, src_scale=src_scale)[0]


class _DualLinear(torch.autograd.Function):
    """z = x Wa^T + y Wb^T: the second GEMM accumulates into the first one's output (beta = 1), so GraphSage's
    `self_linear(x) + neigh_linear(agg)` (pgl/nn/conv.py:104-109) has no separate add pass; both biases go to the epilogue."""

    @staticmethod
    def forward(ctx, x, y, wa, wb):
        ctx.save_for_backward(x, y, wa, wb)
        z = torch.nn.functional.linear(x, wa)
        return z.addmm_(y, wb.t())

    @staticmethod
    def backward(ctx, g):
        x, y, wa, wb = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ wa if ctx.needs_input_grad[0] else None
        gy = g @ wb if ctx.needs_input_grad[1] else None
        gwa = _tall_wgrad(g, x) if ctx.needs_input_grad[2] else None
        gwb = _tall_wgrad(g, y) if ctx.needs_input_grad[3] else None
        return gx, gy, gwa, gwb


def dual_linear(x, y, wa, wb):
    return _DualLinear.apply(x, y, wa, wb)


class _AggregateDualLinear(torch.autograd.Function):
    """z = x Wa^T + (A x) Wb^T -- GraphSageConv on ONE feature tensor (full-graph mode: `feature` is both the source and the destination
    side, pgl/nn/conv.py:99-109) as one autograd node.  Forward is what send_recv + _DualLinear do.  Backward: x receives a gradient from
    both branches, d x = g Wa + A^T (g Wb); as two nodes autograd adds them with an element pass over [N, d] (0.30 ms at N = 2^20,
    d = 128: 3 % of the example model's training step); here the transposed aggregation ACCUMULATES into the GEMM's output
    (pglamd_aggregate, accumulate = 1).  sum / mean."""

    @staticmethod
    def forward(ctx, x, wa, wb, csr, csr_t, rop):
        nb = ops.aggregate(x, csr, rop, int(x.shape[0]))
        ctx.csr, ctx.csr_t, ctx.rop = csr, csr_t, rop
        ctx.save_for_backward(x, nb, wa, wb)
        z = torch.nn.functional.linear(x, wa)
        return z.addmm_(nb, wb.t())

    @staticmethod
    def backward(ctx, g):
        x, nb, wa, wb = ctx.saved_tensors
        g = g.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = g @ wa
            gnb = g @ wb
            scale = None
            if ctx.rop == "mean":                        # d out[v] / d msg = 1 / indeg(v): the per-source scale of the transposed sum
                scale = 1.0 / ctx.csr.degree.clamp(min=1).to(torch.float32)
            ops.aggregate(gnb, ctx.csr_t(), "sum", int(x.shape[0]), src_scale=scale, out=gx, accumulate=1)
        gwa = _tall_wgrad(g, x) if ctx.needs_input_grad[1] else None
        gwb = _tall_wgrad(g, nb) if ctx.needs_input_grad[2] else None
        return gx, gwa, gwb, None, None, None


def aggregate_dual_linear(x, wa, wb, csr, csr_t, rop):
    return _AggregateDualLinear.apply(x, wa, wb, csr, csr_t, rop)
