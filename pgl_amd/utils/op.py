"""Mirror of pgl/utils/op.py: read_rows / RowReader / get_index_from_counts / all_reduce_sum_with_grad."""
import numpy as np
import torch

from .. import autograd as _ag


def read_rows(data, index):
    """pgl/utils/op.py:24-45: row gather over a tensor or a (nested) dict of tensors."""
    if data is None:
        return None
    if isinstance(data, dict):
        return {k: read_rows(v, index) for k, v in data.items()}
    return _ag.gather_rows(data, index)


def get_index_from_counts(counts):
    """pgl/utils/op.py:48-72: exclusive prefix sum with the total appended."""
    if isinstance(counts, torch.Tensor):
        return torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)])
    index = np.cumsum(counts, dtype="int64")
    return np.insert(index, 0, 0)


class RowReader(dict):
    """pgl/utils/op.py:75-87: gathers a key's rows lazily, once."""

    def __init__(self, nfeat, index):
        super(RowReader, self).__init__()
        self.nfeat = nfeat
        self.loaded_nfeat = {}
        self.index = index

    def __getitem__(self, key):
        if key not in self.loaded_nfeat:
            val = self.nfeat[key]
            if type(val).__name__ == "EdgeTensor":          # (pgl_amd/edge_tensor.py: read in original edge order)
                val = val.materialize()
            self.loaded_nfeat[key] = read_rows(val, self.index)
        return self.loaded_nfeat[key]

    def __contains__(self, key):
        return key in self.nfeat

    def keys(self):
        return self.nfeat.keys()


class _AllReduceSum(torch.autograd.Function):
    """Sum over the ranks of the default process group; the gradient of that sum w.r.t. each rank's contribution is the
    sum of the ranks' output gradients (what Paddle's c_allreduce_sum registers as its backward)."""

    @staticmethod
    def forward(ctx, tensor, group):
        import torch.distributed as dist
        ctx.group = group
        out = tensor.contiguous().clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, grad):
        import torch.distributed as dist
        g = grad.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def all_reduce_sum_with_grad(tensor, group=0):
    """pgl/utils/op.py:90-122 (the collective behind the reference's DistGPUGraph): all-reduce(sum) that autograd can
    differentiate.  Out of place; a single process (no initialised group) returns the tensor unchanged.
    `DistGraph` (pgl_amd/distributed.py) does not need it -- its layout has no reduction
    collective -- it is here for code written against the reference's edge-sharded scheme."""
    import torch.distributed as dist
    if group == 0:                                   # Paddle's ring id 0 = the default group (the reference's default argument)
        group = None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensor
    return _AllReduceSum.apply(tensor, group)
