"""paddle.geometric.* answered by the oracle's restatement (oracle/ref_ops.py) -- see that file for the Paddle
op contracts each function follows.  Autograd is provided for the float ops through the same restatement on the
reversed edges, so the reference's training loops can be driven too."""
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


def send_u_recv(x, src_index, dst_index, reduce_op="sum", out_size=None, name=None):
    out = R.c_send_u_recv(_np(x), _np(src_index), _np(dst_index), reduce_op.lower(), _out_size(out_size))
    return torch.as_tensor(out)


def send_ue_recv(x, y, src_index, dst_index, message_op="add", reduce_op="sum", out_size=None, name=None):
    out = R.c_send_ue_recv(_np(x), _np(y), _np(src_index), _np(dst_index), message_op.lower(), reduce_op.lower(),
                           _out_size(out_size))
    return torch.as_tensor(out)


def send_uv(x, y, src_index, dst_index, message_op="add", name=None):
    return torch.as_tensor(R.c_send_uv(_np(x), _np(y), _np(src_index), _np(dst_index), message_op.lower()))


def _segment(op):
    def f(data, segment_ids, name=None):
        return torch.as_tensor(R.c_segment(_np(data), _np(segment_ids), op))
    return f


segment_sum, segment_mean, segment_max, segment_min = (_segment(o) for o in ("sum", "mean", "max", "min"))


def sample_neighbors(*a, **k):
    raise NotImplementedError("sampling is not part of the oracle's paddle stand-in")


def reindex_graph(*a, **k):
    raise NotImplementedError("sampling is not part of the oracle's paddle stand-in")
