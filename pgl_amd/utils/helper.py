"""Mirror of the hot-path helpers in pgl/utils/helper.py (reference lines cited per function)."""
import os

import numpy as np
import torch

from .. import ops


def check_is_tensor(*data):
    """pgl/utils/helper.py check_is_tensor: True if any argument is a device tensor."""
    return any(isinstance(d, torch.Tensor) for d in data)


def default_device():
    """The device Graph.tensor() targets: this rank's MI355X.  Raises when no GPU is visible --
    tensor-mode graphs exist only on the accelerator (no CPU fallback)."""
    if not torch.cuda.is_available():
        raise RuntimeError("pgl_amd: Graph.tensor() needs an MI355X (torch.cuda.is_available() is False); "
                           "tensor-mode message passing has no CPU fallback")
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())


def to_device_tensor(data, device=None):
    """pgl/utils/helper.py:32-43 to_paddle_tensor (UVA mode is not reproduced: 288 GB of HBM per
    GPU holds the structures the reference had to leave in host memory)."""
    if isinstance(data, torch.Tensor):
        return data if device is None or data.device == device else data.to(device)
    device = device or default_device()
    return torch.as_tensor(np.ascontiguousarray(data)).to(device)


def maybe_num_nodes(edges):
    """pgl/utils/helper.py:133-153."""
    if isinstance(edges, torch.Tensor):
        return int(edges.max().item()) + 1 if edges.numel() else 0
    if len(edges) == 0:
        return 0
    return int(np.max(edges)) + 1


def unique_segment(data, dtype="int64"):
    """pgl/utils/helper.py:156-160 for arbitrary SORTED device keys (paddle.unique on a sorted
    array == run-length ranks).  Graph.get_segment_ids uses the cached-CSR fast path instead."""
    if not isinstance(data, torch.Tensor):
        uniq, inv = np.unique(np.asarray(data), return_inverse=True)
        return uniq.astype(dtype), inv.reshape(-1).astype(dtype)
    n = int(data.shape[0])
    if n == 0:
        return data.new_zeros(0, dtype=torch.int64), data.new_zeros(0, dtype=torch.int64)
    hi = int(data[-1].item()) + 1
    seg_ptr = ops.seg_ptr_from_ids(data, hi)
    degree = seg_ptr[1:] - seg_ptr[:-1]
    return ops.unique_segment(degree, data.to(torch.int64))


def generate_segment_id_from_index(index):
    """pgl/utils/helper.py:116-130 (numpy path; used for graph_node_id of batched graphs)."""
    index = np.asarray(index)
    seg = np.zeros(int(index[-1]) + 1, dtype="int32")
    np.add.at(seg, index[:-1], 1)
    return (np.cumsum(seg)[:-1] - 1).astype("int32")


def scatter(x, index, updates, overwrite=True, name=None):
    """pgl/utils/helper.py:46-117 (paddle.scatter): rows of `x` named by `index` replaced by the rows of `updates`
    (overwrite=True; with duplicate ids the LAST update wins here, the reference leaves the order unspecified) or first
    zeroed and then summed over the duplicates (overwrite=False).  Out of place."""
    index = torch.as_tensor(index, device=x.device).reshape(-1).long()
    if overwrite:
        out = x.clone()
        # index_copy_ with duplicates is order-dependent on the device: resolve to the last occurrence explicitly
        last = torch.full((x.shape[0],), -1, dtype=torch.int64, device=x.device)
        last.scatter_reduce_(0, index, torch.arange(index.shape[0], device=x.device), "amax", include_self=True)
        hit = last >= 0
        out[hit] = updates[last[hit]].to(x.dtype)
        return out
    out = x.clone()
    out[index] = 0
    return out.index_add(0, index, updates.to(x.dtype))


def to_paddle_tensor(data, uva=False):
    """pgl/utils/helper.py:32-43: a numpy array as a device tensor.  `uva` (pinned host memory mapped into the GPU's address space, the
    reference's capacity workaround) is accepted and means HBM here; like the reference it refuses uva without a GPU."""
    if uva and not torch.cuda.is_available():
        raise ValueError("UVA tensor should be used under GPU environment.")
    return to_device_tensor(data)


def graph_send_recv(x, src_index, dst_index, pool_type="sum"):
    """pgl/utils/helper.py:163-210 -- the reference's own fallback for `send_recv` on RAW index arrays (an [E, d] gather followed by a
    scatter-add into zeros([N, d])), kept by it for Paddle versions without `paddle.geometric`.  Here it is the engine's raw-COO entry
    (`ops.send_u_recv`: the edge-parallel atomic kernel for small fp32 sums, csr_build + the flat kernel otherwise -- DESIGN section 3 K1');
    no [E, d] message tensor exists.  The reference implements "sum" only and asserts so; the other three pool types work here."""
    assert pool_type in ("sum", "mean", "max", "min"), "pool_type must be one of 'sum', 'mean', 'max', 'min'"
    return ops.send_u_recv(x, src_index, dst_index, pool_type)

