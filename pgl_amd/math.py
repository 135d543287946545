"""pgl_amd.math -- segment operators.  Mirrors pgl/math.py:30-364.

`num_segments` is an optional extension: when the caller knows ids[-1]+1 it avoids a device sync.
"""
import torch

from . import autograd as ag
from . import ops

__all__ = ["segment_pool", "segment_sum", "segment_mean", "segment_max", "segment_min", "segment_softmax",
           "segment_padding", "segment_topk"]


def segment_pool(data, segment_ids, pool_type, name=None, num_segments=None):
    """pgl/math.py:30-47."""
    pool_type = pool_type.upper()
    if pool_type not in ("SUM", "MEAN", "MAX", "MIN"):
        raise ValueError("We only support sum, mean, max, min pool types in segment_pool function.")
    return ag.segment_reduce(data, segment_ids, pool_type.lower(), num_segments)


def segment_sum(data, segment_ids, name=None, num_segments=None):
    """pgl/math.py:49-79."""
    return ag.segment_reduce(data, segment_ids, "sum", num_segments)


def segment_mean(data, segment_ids, name=None, num_segments=None):
    """pgl/math.py:82-113."""
    return ag.segment_reduce(data, segment_ids, "mean", num_segments)


def segment_min(data, segment_ids, name=None, num_segments=None):
    """pgl/math.py:116-145."""
    return ag.segment_reduce(data, segment_ids, "min", num_segments)


def segment_max(data, segment_ids, name=None, num_segments=None):
    """pgl/math.py:148-178."""
    return ag.segment_reduce(data, segment_ids, "max", num_segments)


def segment_softmax(data, segment_ids, num_segments=None):
    """pgl/math.py:181-224: max / exp / sum / div as four balanced launches instead of 7 tensor ops."""
    if num_segments is None:
        num_segments = int(segment_ids[-1].item()) + 1 if int(segment_ids.shape[0]) else 0
    seg_ptr = ops.seg_ptr_from_ids(segment_ids, num_segments)
    ids32 = segment_ids if segment_ids.dtype == torch.int32 else ops.narrow_i64(segment_ids)
    return ag.segment_softmax(data, ops.SegView(seg_ptr, ids32, ids32, None))


def segment_padding(data, segment_ids):
    """pgl/math.py:227-272: segments laid out as rows of a zero-padded [num_segments, max_len, dim] tensor.
    -> (output, segment_len [num_segments], index [n, 2] = (segment, position inside it))."""
    ids = segment_ids.long()
    n_seg = int(ids[-1].item()) + 1 if int(ids.shape[0]) else 0
    seg_len = torch.bincount(ids, minlength=n_seg)
    max_len = int(seg_len.max().item()) if n_seg else 0
    first = torch.cumsum(seg_len, 0) - seg_len
    pos = torch.arange(ids.shape[0], device=ids.device) - first[ids]
    out = torch.zeros((n_seg, max_len, data.shape[-1]), dtype=data.dtype, device=data.device)
    out = out.index_put((ids, pos), data)
    return out, seg_len, torch.stack([ids, pos], dim=1)


def segment_topk(x, scores, segment_ids, ratio, min_score=None, return_index=False):
    """pgl/math.py:299-364: per segment, the rows with the k highest scores (k = ceil(ratio * len), or min(ratio, len) for an
    integer ratio), in descending score order, segments in order; with `min_score` every row scoring above
    min(min_score, segment max - 1e-7) is kept instead, in input order."""
    ids = segment_ids.long()
    scores = scores.reshape(-1)
    if min_score is not None:
        seg_max = segment_max(scores.reshape(-1, 1), segment_ids).reshape(-1)[ids] - 1e-7
        perm = (scores > seg_max.clamp(max=min_score)).nonzero().reshape(-1)
    else:
        n_seg = int(ids[-1].item()) + 1 if int(ids.shape[0]) else 0
        seg_len = torch.bincount(ids, minlength=n_seg)
        max_len = int(seg_len.max().item()) if n_seg else 0
        first = torch.cumsum(seg_len, 0) - seg_len
        pos = torch.arange(ids.shape[0], device=ids.device) - first[ids]
        dense = torch.full((n_seg, max_len), -1e20, dtype=torch.float32, device=scores.device)
        dense[ids, pos] = scores.detach().float()
        order = torch.argsort(dense, dim=1, descending=True, stable=True) + first.reshape(-1, 1)
        if isinstance(ratio, int):
            k = torch.clamp(seg_len, max=ratio)
        else:
            k = torch.ceil(ratio * seg_len.float()).long()
        take = torch.arange(max_len, device=ids.device).reshape(1, -1) < k.reshape(-1, 1)
        perm = order[take]
    out = x[perm]
    return (out, perm) if return_index else out
