"""EdgeIndex -- CSR/CSC view of the edge list.  Mirrors pgl/utils/edge_index.py (reference).

numpy mode  -> host counting sort in libpglamd (pglamd_build_index_host), as the reference's
               numpy path calls graph_kernel.build_index (edge_index.py:56-57);
tensor mode -> HIP radix-sort CSR build (pglamd_csr_build) instead of the reference's
               scatter + argsort + gather + cumsum (edge_index.py:42-54), and bit-identical to
               the numpy path (the reference's argsort path does not guarantee that).
"""
import os

import numpy as np
import torch

from .. import ops
from .helper import check_is_tensor, to_device_tensor


class EdgeIndex(object):
    def __init__(self):
        self._degree = self._indptr = None
        self._sv64 = self._su64 = self._se64 = None
        self._is_tensor = False
        self._csr = None          # ops.CSR with the int32 engine copies (tensor mode only)

    # The reference's three int64 [E] arrays.  In tensor mode the engine builds and reads int32 copies only; the int64
    # views the reference API exposes (sorted_edges, dump, view_v ...) are widened from them on first access.
    def _widened(self, held, name32):
        if held is None and self._csr is not None:
            held = getattr(self._csr, name32).to(torch.int64)
        return held

    @property
    def _sorted_v(self):
        self._sv64 = self._widened(self._sv64, "col32")
        return self._sv64

    @_sorted_v.setter
    def _sorted_v(self, value):
        self._sv64 = value

    @property
    def _sorted_u(self):
        self._su64 = self._widened(self._su64, "row32")
        return self._su64

    @_sorted_u.setter
    def _sorted_u(self, value):
        self._su64 = value

    @property
    def _sorted_eid(self):
        self._se64 = self._widened(self._se64, "eid32")
        return self._se64

    @_sorted_eid.setter
    def _sorted_eid(self, value):
        self._se64 = value

    # ---- construction ------------------------------------------------------------------------
    @classmethod
    def from_edges(cls, u, v, num_nodes, check_range=True):
        self = cls()
        self._is_tensor = check_is_tensor(u, v)
        if self._is_tensor:
            c = ops.csr_build(u, v, int(num_nodes), want_i64=False, check_range=check_range)
            self._adopt(c)
        else:
            self._degree, self._sorted_v, self._sorted_u, self._sorted_eid, self._indptr = \
                ops.host_build_index(u, v, int(num_nodes))
        return self

    @classmethod
    def from_sorted(cls, u, v, num_nodes):
        """Tensor-mode index of edges whose keys `u` are ALREADY non-decreasing (sampled blocks): no sort, the caller vouches
        for the order.  Same arrays as from_edges(u, v) would produce."""
        self = cls()
        self._is_tensor = True
        self._adopt(ops.csr_from_sorted(u, v, int(num_nodes)))
        return self

    @classmethod
    def from_index(cls, sorted_v, sorted_u, sorted_eid, degree, indptr):
        self = cls()
        self._degree, self._sorted_v, self._sorted_u = degree, sorted_v, sorted_u
        self._sorted_eid, self._indptr = sorted_eid, indptr
        self._is_tensor = check_is_tensor(sorted_v, sorted_u, sorted_eid, degree, indptr)
        if self._is_tensor:
            self._make_engine_copies()
        return self

    def _adopt(self, c):
        self._csr = c
        self._degree, self._indptr = c.degree, c.indptr
        self._sv64, self._su64, self._se64 = c.sorted_v, c.sorted_u, c.sorted_eid      # None when built without them

    def _make_engine_copies(self):
        c = ops.CSR()
        c.degree, c.sorted_v, c.sorted_u = self._degree, self._sorted_v, self._sorted_u
        c.sorted_eid, c.indptr = self._sorted_eid, self._indptr
        c.num_nodes, c.num_edges = int(self._degree.shape[0]), int(self._sorted_u.shape[0])
        c.row32 = ops.narrow_i64(self._sorted_u)
        c.col32 = ops.narrow_i64(self._sorted_v)
        c.eid32 = ops.narrow_i64(self._sorted_eid)
        self._csr = c

    # ---- accessors (reference API) -----------------------------------------------------------
    @property
    def degree(self):
        return self._degree

    @property
    def csr(self):
        """Engine view (int32 row/col/eid + int64 indptr/degree).  Tensor mode only."""
        if not self._is_tensor:
            raise ValueError("EdgeIndex.csr needs tensor mode; call tensor() first")
        return self._csr

    def view_v(self, u=None):
        if self._is_tensor:
            raise NotImplementedError("not implemented!")
        if u is None:
            return np.split(self._sorted_v, self._indptr[1:-1])
        u = np.array(u, dtype="int64")
        return np.array([self._sorted_v[self._indptr[j]:self._indptr[j + 1]] for j in u], dtype=object)

    def view_eid(self, u=None):
        if self._is_tensor:
            raise NotImplementedError("not implemented!")
        if u is None:
            return np.split(self._sorted_eid, self._indptr[1:-1])
        u = np.array(u, dtype="int64")
        return np.array([self._sorted_eid[self._indptr[j]:self._indptr[j + 1]] for j in u], dtype=object)

    def triples(self):
        return self._sorted_u, self._sorted_v, self._sorted_eid

    def is_tensor(self):
        return self._is_tensor

    # ---- conversion --------------------------------------------------------------------------
    def tensor(self, inplace=True, uva=False, device=None):
        """pgl/utils/edge_index.py:139-171.  `uva` is the reference's SECOND positional argument (index kept in pinned host memory,
        a capacity workaround): accepted, the index goes to HBM either way (Graph.tensor says why)."""
        if uva and not torch.cuda.is_available():
            raise ValueError("uva tensor graph should be run under gpu environment!")
        if self._is_tensor:
            return self
        arrs = [to_device_tensor(a, device) for a in
                (self._sorted_v, self._sorted_u, self._sorted_eid, self._degree, self._indptr)]
        if inplace:
            self._sorted_v, self._sorted_u, self._sorted_eid, self._degree, self._indptr = arrs
            self._is_tensor = True
            self._make_engine_copies()
            return self
        return EdgeIndex.from_index(sorted_v=arrs[0], sorted_u=arrs[1], sorted_eid=arrs[2], degree=arrs[3],
                                    indptr=arrs[4])

    def numpy(self, inplace=True):
        if not self._is_tensor:
            return self
        arrs = [a.cpu().numpy() for a in
                (self._sorted_v, self._sorted_u, self._sorted_eid, self._degree, self._indptr)]
        if inplace:
            self._sorted_v, self._sorted_u, self._sorted_eid, self._degree, self._indptr = arrs
            self._is_tensor = False
            self._csr = None
            return self
        return EdgeIndex.from_index(sorted_v=arrs[0], sorted_u=arrs[1], sorted_eid=arrs[2], degree=arrs[3],
                                    indptr=arrs[4])

    # ---- on-disk format: five int64 .npy files, identical to the reference (edge_index.py:72-95,208-219)
    @classmethod
    def load(cls, path, mmap_mode="r"):
        self = cls()
        for name in ("degree", "sorted_u", "sorted_v", "sorted_eid", "indptr"):
            setattr(self, "_" + name, np.load(os.path.join(path, name + ".npy"), mmap_mode=mmap_mode))
        self._is_tensor = False
        return self

    def dump(self, path):
        if self._is_tensor:
            return self.numpy(inplace=False).dump(path)
        os.makedirs(path, exist_ok=True)
        for name in ("degree", "sorted_v", "sorted_u", "sorted_eid", "indptr"):
            np.save(os.path.join(path, name + ".npy"), getattr(self, "_" + name))
