"""Mirror of pgl/utils/op.py: read_rows / RowReader / get_index_from_counts."""
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
            self.loaded_nfeat[key] = read_rows(self.nfeat[key], self.index)
        return self.loaded_nfeat[key]

    def __contains__(self, key):
        return key in self.nfeat

    def keys(self):
        return self.nfeat.keys()
