"""pgl_amd.math -- segment operators.  Mirrors pgl/math.py:30-224 (segment_padding / segment_topk
are graph-pooling helpers outside the message-passing path and are not provided).

`num_segments` is an optional extension: when the caller knows ids[-1]+1 it avoids a device sync.
"""
import torch

from . import autograd as ag
from . import ops

__all__ = ["segment_pool", "segment_sum", "segment_mean", "segment_max", "segment_min", "segment_softmax"]


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
