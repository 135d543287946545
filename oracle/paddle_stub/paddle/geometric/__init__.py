"""paddle.geometric.* answered by the oracle's restatement (oracle/ref_ops.py) -- see that file for the Paddle
op contracts each function follows.

Forward values always come from the C restatement.  When an input requires grad, the op is wrapped in an
autograd.Function whose backward differentiates an independent plain-torch formulation of the same op (index_add_ /
scatter_reduce), so the reference's layers can also produce GRADIENT fixtures (tests/golden/make_golden_layers.py).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import ref_ops as R  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _out_size(o):
    if o is None:
        return None
    if isinstance(o, torch.Tensor):
        o = int(o.reshape(-1)[0].item())
    return int(o) if int(o) > 0 else None


def _i64(t):
    return torch.as_tensor(_np(t), dtype=torch.int64).reshape(-1)


# ---- plain-torch formulations (used for backward only) --------------------------------------------------
def _reduce_rows(msg, dst, m, op):
    shape = (m,) + tuple(msg.shape[1:])
    if op in ("sum", "mean"):
        out = torch.zeros(shape, dtype=msg.dtype).index_add_(0, dst, msg)
        if op == "mean":
            cnt = torch.zeros(m, dtype=msg.dtype).index_add_(0, dst, torch.ones(len(dst), dtype=msg.dtype)).clamp(min=1)
            out = out / cnt.reshape((m,) + (1,) * (msg.dim() - 1))
        return out
    # max / min: Paddle's gradient rule gives the FULL output gradient to every message equal to the winner (its grad
    # kernels accumulate `grad * (x == out)` per edge; torch's scatter_reduce would split it among ties), so the
    # backward formulation is a surrogate whose derivative w.r.t. msg is exactly that mask
    idx = dst.reshape((-1,) + (1,) * (msg.dim() - 1)).expand_as(msg)
    win = torch.zeros(shape, dtype=msg.dtype).scatter_reduce(0, idx, msg.detach(), "amax" if op == "max" else "amin", include_self=False)
    mask = (msg.detach() == win[dst]).to(msg.dtype)
    return torch.zeros(shape, dtype=msg.dtype).index_add_(0, dst, msg * mask)


_MOP = {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div}


def _t_send_u_recv(x, src, dst, op, m):
    return _reduce_rows(x[src], dst, m, op)


def _t_send_ue_recv(x, y, src, dst, mop, rop, m):
    return _reduce_rows(_MOP[mop](x[src], y), dst, m, rop)


def _t_send_uv(x, y, src, dst, mop):
    return _MOP[mop](x[src], y[dst])


def _t_segment(data, ids, op):
    m = int(ids[-1]) + 1 if len(ids) else 0
    return _reduce_rows(data, ids, m, op)


class _ViaOracle(torch.autograd.Function):
    """forward: value computed by the C restatement; backward: autograd of the torch formulation `fn(*tensors)`."""

    @staticmethod
    def forward(ctx, fn, value, *tensors):
        ctx.fn = fn
        ctx.save_for_backward(*tensors)
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(t.is_floating_point()) for t in ctx.saved_tensors]
            out = ctx.fn(*ins)
            need = [t for t in ins if t.requires_grad]
            grads = torch.autograd.grad(out, need, g, allow_unused=True)
        it = iter(grads)
        return (None, None) + tuple(next(it) if t.requires_grad else None for t in ins)


def _wrap(fn, value, *tensors):
    value = torch.as_tensor(value)
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        return _ViaOracle.apply(fn, value, *tensors)
    return value


def send_u_recv(x, src_index, dst_index, reduce_op="sum", out_size=None, name=None):
    op = reduce_op.lower(); o = _out_size(out_size)
    val = R.c_send_u_recv(_np(x), _np(src_index), _np(dst_index), op, o)
    src, dst, m = _i64(src_index), _i64(dst_index), val.shape[0]
    return _wrap(lambda xx: _t_send_u_recv(xx, src, dst, op, m), val, x)


def send_ue_recv(x, y, src_index, dst_index, message_op="add", reduce_op="sum", out_size=None, name=None):
    mop, rop, o = message_op.lower(), reduce_op.lower(), _out_size(out_size)
    val = R.c_send_ue_recv(_np(x), _np(y), _np(src_index), _np(dst_index), mop, rop, o)
    src, dst, m = _i64(src_index), _i64(dst_index), val.shape[0]
    return _wrap(lambda xx, yy: _t_send_ue_recv(xx, yy, src, dst, mop, rop, m), val, x, y)


def send_uv(x, y, src_index, dst_index, message_op="add", name=None):
    mop = message_op.lower()
    val = R.c_send_uv(_np(x), _np(y), _np(src_index), _np(dst_index), mop)
    src, dst = _i64(src_index), _i64(dst_index)
    return _wrap(lambda xx, yy: _t_send_uv(xx, yy, src, dst, mop), val, x, y)


def _segment(op):
    def f(data, segment_ids, name=None):
        val = R.c_segment(_np(data), _np(segment_ids), op)
        ids = _i64(segment_ids)
        return _wrap(lambda dd: _t_segment(dd, ids, op), val, data)
    return f


segment_sum, segment_mean, segment_max, segment_min = (_segment(o) for o in ("sum", "mean", "max", "min"))


def sample_neighbors(*a, **k):
    raise NotImplementedError("sampling is not part of the oracle's paddle stand-in")


def reindex_graph(*a, **k):
    raise NotImplementedError("sampling is not part of the oracle's paddle stand-in")
