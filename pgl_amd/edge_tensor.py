"""EdgeTensor -- [E, ...] tensors that stay in the ENGINE'S edge order (destination-sorted) across an op chain.

The reference's un-fused attention chains (GATConv pgl/nn/conv.py:331-339, GATv2Conv :421-424, TransformerConv :796-834, FAConv,
any user composition of Graph.send_uv / GF.edge_softmax / Graph.send_ue_recv) keep every [E, H] tensor in ORIGINAL edge order.
The kernels walk the destination-sorted edge stream, so each of those ops reaches its rows through the eid permutation: one
128-byte line per 32-byte row at H = 8 -- measured at C3 (profiles/r04/edgeops.txt) send_uv 0.52 ms and edge_softmax 1.01 ms in
original order against 0.31 / 0.76 ms when the rows already lie in sorted order.

So `Graph.send_uv` / `Graph.sddmm` hand back an EdgeTensor: the rows physically in destination-sorted order plus the view that
knows the permutation.  Element-wise work (leaky_relu, tanh, dropout, * attn, sum over the trailing dims, reshape of the
trailing dims, ...) is applied to the sorted rows and keeps the tag; `GF.edge_softmax(graph, t)` and
`Graph.send_ue_recv(x, t, ...)` consume the sorted rows directly (no eid indirection at all).  The moment anything else touches
the data -- indexing, .cpu(), a reduction over the edge dimension, an op this file does not list -- the rows are permuted back
to ORIGINAL edge order (one differentiable gather, cached) and the op runs on an ordinary tensor: what a caller READS is always
what the reference would have produced, in the reference's order (pgl/nn/functional/graph_op.py:117-123).

EdgeTensor is deliberately NOT a torch.Tensor subclass: the engine's ops hand raw device pointers to libpglamd, and a subclass
whose storage is in another row order than it claims would be read wrongly by any op that does not know about it.  As a plain
wrapper it cannot reach a kernel by accident (it has no data_ptr); everything goes through `__torch_function__`, the arithmetic
dunders below, or `__getattr__` (which materialises first).  PGLAMD_EDGE_TENSOR=0 (or graph.lazy_edge_order = False) turns the
mechanism off: send_uv then returns ordinary original-order tensors as before.
"""
import os

import torch
import torch.nn.functional as F

ENABLED = os.environ.get("PGLAMD_EDGE_TENSOR", "1") != "0"

# ops that act element by element (or only on trailing dimensions) and therefore commute with a permutation of the rows
_UNARY = {torch.exp, torch.tanh, torch.sigmoid, torch.relu, torch.abs, torch.neg, torch.log, torch.sqrt, torch.rsqrt, torch.square,
          torch.clamp, torch.clamp_min, torch.clamp_max, torch.clone, torch.detach, torch.nan_to_num, torch.erf, torch.reciprocal,
          F.leaky_relu, F.relu, F.elu, F.gelu, F.silu, F.sigmoid, F.tanh, F.dropout, F.softplus, F.hardtanh, F.selu, F.relu6,
          torch.Tensor.exp, torch.Tensor.tanh, torch.Tensor.sigmoid, torch.Tensor.relu, torch.Tensor.abs, torch.Tensor.neg,
          torch.Tensor.clamp, torch.Tensor.float, torch.Tensor.half, torch.Tensor.double, torch.Tensor.bfloat16, torch.Tensor.detach,
          torch.Tensor.clone, torch.Tensor.contiguous, torch.Tensor.square, torch.Tensor.sqrt, torch.Tensor.log}
_BINARY = {torch.add, torch.sub, torch.mul, torch.div, torch.true_divide, torch.maximum, torch.minimum, torch.pow,
           torch.Tensor.add, torch.Tensor.sub, torch.Tensor.mul, torch.Tensor.div, torch.Tensor.true_divide, torch.Tensor.pow,
           torch.Tensor.__add__, torch.Tensor.__radd__, torch.Tensor.__sub__, torch.Tensor.__rsub__, torch.Tensor.__mul__,
           torch.Tensor.__rmul__, torch.Tensor.__truediv__, torch.Tensor.__rtruediv__, torch.Tensor.__pow__}
_REDUCE_TRAILING = {torch.sum, torch.mean, torch.amax, torch.amin, torch.Tensor.sum, torch.Tensor.mean, torch.Tensor.amax, torch.Tensor.amin}


def _extras_safe(extra_args, kwargs, ref):
    """True when every tensor among the EXTRA operands of an element-wise call (torch.clamp(t, min=<tensor>), alpha=..., ...) can be
    combined with the sorted rows `ref` without knowing the edge order (an [E, ...] tensor in original order cannot)."""
    for v in list(extra_args) + list(kwargs.values()):
        if isinstance(v, EdgeTensor) or (isinstance(v, torch.Tensor) and not _rows_safe(v, ref)):
            return False
    return "out" not in kwargs


def _rows_safe(other, ref):
    """True when `other` can be combined element-wise with rows `ref` ([E, ...]) without knowing the edge order: a Python scalar,
    or a tensor that broadcasts over the edge dimension (fewer dims, or size 1 there)."""
    if isinstance(other, (int, float, bool)):
        return True
    if isinstance(other, torch.Tensor):
        if other.dim() < ref.dim():
            return True
        return other.dim() == ref.dim() and int(other.shape[0]) == 1 and int(ref.shape[0]) != 1
    return False


class EdgeTensor(object):
    __slots__ = ("_v", "_view", "_orig")

    def __init__(self, sorted_rows, view):
        self._v, self._view, self._orig = sorted_rows, view, None

    # ---- what identifies it ------------------------------------------------------------------------------------------------
    @property
    def graph(self):
        return self._view.graph

    def sorted_rows(self, graph=None):
        """The rows in destination-sorted order if this tensor belongs to `graph` (None: any), else None."""
        return self._v if (graph is None or self._view.graph is graph) else None

    def materialize(self):
        """An ordinary tensor in ORIGINAL edge order (differentiable; computed once)."""
        if self._orig is None:
            self._orig = self._view.from_order(self._v)
        return self._orig

    # ---- tensor facts that do not depend on the row order ----------------------------------------------------------------------
    shape = property(lambda self: self._v.shape)
    dtype = property(lambda self: self._v.dtype)
    device = property(lambda self: self._v.device)
    ndim = property(lambda self: self._v.dim())
    is_cuda = property(lambda self: self._v.is_cuda)
    requires_grad = property(lambda self: self._v.requires_grad)

    def dim(self):
        return self._v.dim()

    def size(self, *a):
        return self._v.size(*a)

    def numel(self):
        return self._v.numel()

    def __len__(self):
        return int(self._v.shape[0])

    def _wrap(self, rows):
        return EdgeTensor(rows, self._view)

    # ---- shape changes that leave the edge dimension alone ------------------------------------------------------------------------
    def _keeps_rows(self, new):
        return new.dim() >= 1 and int(new.shape[0]) == int(self._v.shape[0])

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        E = int(self._v.shape[0])
        if len(shape) >= 1 and shape[0] in (-1, E) and (shape[0] == E or (self._v.numel() // max(E, 1)) == _known_prod(shape[1:])):
            return self._wrap(self._v.reshape((E,) + tuple(shape[1:])))
        return self.materialize().reshape(*shape)

    view = reshape

    def unsqueeze(self, dim):
        d = dim if dim >= 0 else dim + self._v.dim() + 1
        return self._wrap(self._v.unsqueeze(dim)) if d != 0 else self.materialize().unsqueeze(dim)

    def squeeze(self, dim=None):
        if dim is None or (dim if dim >= 0 else dim + self._v.dim()) == 0:
            return self.materialize().squeeze() if dim is None else self.materialize().squeeze(dim)
        return self._wrap(self._v.squeeze(dim))

    def flatten(self, start_dim=0, end_dim=-1):
        return self._wrap(self._v.flatten(start_dim, end_dim)) if start_dim >= 1 else self.materialize().flatten(start_dim, end_dim)

    def to(self, *a, **k):
        return self._wrap(self._v.to(*a, **k)) if _only_dtype(a, k) else self.materialize().to(*a, **k)

    def float(self):
        return self._wrap(self._v.float())

    def half(self):
        return self._wrap(self._v.half())

    def double(self):
        return self._wrap(self._v.double())

    def contiguous(self):
        return self._wrap(self._v.contiguous())

    def detach(self):
        return self._wrap(self._v.detach())

    def clone(self):
        return self._wrap(self._v.clone())

    def sum(self, dim=None, keepdim=False, **k):
        return self._reduce(torch.sum, dim, keepdim, k)

    def mean(self, dim=None, keepdim=False, **k):
        return self._reduce(torch.mean, dim, keepdim, k)

    def _reduce(self, fn, dim, keepdim, k):
        dims = _dims(dim, self._v.dim())
        if dims is not None and 0 not in dims:
            return self._wrap(fn(self._v, dim=dim, keepdim=keepdim, **k))
        return fn(self.materialize(), **k) if dim is None else fn(self.materialize(), dim=dim, keepdim=keepdim, **k)

    # ---- arithmetic ---------------------------------------------------------------------------------------------------------------
    def _binary(self, fn, other, swap=False):
        if isinstance(other, EdgeTensor):
            if other._view is self._view and other._v.dim() == self._v.dim():
                return self._wrap(fn(other._v, self._v) if swap else fn(self._v, other._v))
            a, b = self.materialize(), other.materialize()
            return fn(b, a) if swap else fn(a, b)
        if _rows_safe(other, self._v):
            return self._wrap(fn(other, self._v) if swap else fn(self._v, other))
        a = self.materialize()
        return fn(other, a) if swap else fn(a, other)

    def __add__(self, o): return self._binary(torch.add, o)
    def __radd__(self, o): return self._binary(torch.add, o, True)
    def __sub__(self, o): return self._binary(torch.sub, o)
    def __rsub__(self, o): return self._binary(torch.sub, o, True)
    def __mul__(self, o): return self._binary(torch.mul, o)
    def __rmul__(self, o): return self._binary(torch.mul, o, True)
    def __truediv__(self, o): return self._binary(torch.div, o)
    def __rtruediv__(self, o): return self._binary(torch.div, o, True)
    def __pow__(self, o): return self._binary(torch.pow, o)
    def __matmul__(self, o):                          # rows @ W: row-wise when the weight is an ordinary (<= 2-D) tensor
        if isinstance(o, torch.Tensor) and o.dim() <= 2:
            return self._wrap(torch.matmul(self._v, o))
        return torch.matmul(self.materialize(), o.materialize() if isinstance(o, EdgeTensor) else o)

    def __neg__(self): return self._wrap(-self._v)
    def __abs__(self): return self._wrap(self._v.abs())

    def exp(self): return self._wrap(self._v.exp())
    def tanh(self): return self._wrap(self._v.tanh())
    def sigmoid(self): return self._wrap(self._v.sigmoid())
    def relu(self): return self._wrap(self._v.relu())
    def abs(self): return self._wrap(self._v.abs())
    def clamp(self, *a, **k):
        return self._wrap(self._v.clamp(*a, **k)) if _extras_safe(a, k, self._v) else self.materialize().clamp(*[materialize(x) for x in a], **{n: materialize(v) for n, v in k.items()})

    # ---- everything else: original order first ------------------------------------------------------------------------------------
    def __getitem__(self, idx):
        return self.materialize()[idx]

    # ---- writes: they land in the original-order copy, and the sorted rows FOLLOW it ---------------------------------
    def _resync(self):
        """The original-order copy was written to in place: the destination-sorted rows are rebuilt from it (one gather), so that
        edge_softmax / send_ue_recv / recv never consume values the caller has since overwritten."""
        self._v = self._view.to_order(self._orig)

    def __setitem__(self, idx, value):
        self.materialize()[materialize(idx)] = materialize(value)
        self._resync()

    def __iter__(self):
        return iter(self.materialize())

    def __repr__(self):
        return "EdgeTensor(in original order: %r)" % (self.materialize(),)

    def __bool__(self):
        return bool(self.materialize())

    def __eq__(self, o): return self.materialize() == (o.materialize() if isinstance(o, EdgeTensor) else o)
    def __ne__(self, o): return self.materialize() != (o.materialize() if isinstance(o, EdgeTensor) else o)
    def __lt__(self, o): return self.materialize() < (o.materialize() if isinstance(o, EdgeTensor) else o)
    def __le__(self, o): return self.materialize() <= (o.materialize() if isinstance(o, EdgeTensor) else o)
    def __gt__(self, o): return self.materialize() > (o.materialize() if isinstance(o, EdgeTensor) else o)
    def __ge__(self, o): return self.materialize() >= (o.materialize() if isinstance(o, EdgeTensor) else o)
    __hash__ = object.__hash__

    def __getattr__(self, name):                      # any Tensor method / attribute not handled above
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        attr = getattr(self.materialize(), name)
        if name.endswith("_") and callable(attr):     # an in-place Tensor method (mul_, clamp_, masked_fill_, zero_, copy_, ...)
            def inplace(*a, **k):
                attr(*[materialize(x) for x in a], **{n: materialize(v) for n, v in k.items()})
                self._resync()
                return self
            return inplace
        return attr

    def __array__(self, *a, **k):
        return self.materialize().detach().cpu().numpy().__array__(*a, **k)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        ets = [a for a in args if isinstance(a, EdgeTensor)]
        first = ets[0] if ets else None
        if first is not None and not any(isinstance(v, EdgeTensor) for v in kwargs.values()):
            if func in _UNARY and isinstance(args[0], EdgeTensor) and len(ets) == 1 and _extras_safe(args[1:], kwargs, first._v):
                return first._wrap(func(first._v, *args[1:], **kwargs))
            if func in _BINARY and len(args) >= 2 and _extras_safe(args[2:], kwargs, first._v):
                a, b = args[0], args[1]
                if isinstance(a, EdgeTensor) and isinstance(b, EdgeTensor):
                    if a._view is b._view and a._v.dim() == b._v.dim():
                        return a._wrap(func(a._v, b._v, *args[2:], **kwargs))
                elif isinstance(a, EdgeTensor) and _rows_safe(b, a._v):
                    return a._wrap(func(a._v, b, *args[2:], **kwargs))
                elif isinstance(b, EdgeTensor) and _rows_safe(a, b._v) and isinstance(a, torch.Tensor):
                    return b._wrap(func(a, b._v, *args[2:], **kwargs))
            if func in _REDUCE_TRAILING and isinstance(args[0], EdgeTensor) and len(ets) == 1:
                dim = kwargs.get("dim", args[1] if len(args) > 1 else None)
                dims = _dims(dim, first._v.dim())
                if dims is not None and 0 not in dims:
                    return first._wrap(func(first._v, *args[1:], **kwargs))
            if func in (torch.reshape, torch.Tensor.reshape, torch.Tensor.view) and isinstance(args[0], EdgeTensor):
                return first.reshape(*args[1:])
            # row-wise ops: every output row depends on the same input row only
            if func in (torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__) and isinstance(args[0], EdgeTensor) and len(ets) == 1 \
                    and isinstance(args[1], torch.Tensor) and args[1].dim() <= 2:
                return first._wrap(func(first._v, args[1]))
            if func is F.linear and isinstance(args[0], EdgeTensor) and len(ets) == 1:
                return first._wrap(func(first._v, *args[1:], **kwargs))
            if func in (F.softmax, F.log_softmax, torch.softmax, torch.log_softmax, F.normalize, F.layer_norm) and isinstance(args[0], EdgeTensor) and len(ets) == 1:
                dim = kwargs.get("dim", args[1] if (len(args) > 1 and isinstance(args[1], int)) else (-1 if func is F.layer_norm else None))
                if func is F.normalize and dim is None:
                    dim = 1
                if dim is not None and (dim if dim >= 0 else dim + first._v.dim()) != 0 and first._v.dim() >= 2:
                    return first._wrap(func(first._v, *args[1:], **kwargs))
        if func in (torch.cat, torch.concat) and args and isinstance(args[0], (list, tuple)) and args[0]:
            parts = list(args[0])
            dim = kwargs.get("dim", kwargs.get("axis", args[1] if len(args) > 1 else 0))
            if all(isinstance(t, EdgeTensor) for t in parts) and all(t._view is parts[0]._view and t._v.dim() == parts[0]._v.dim() for t in parts):
                if (dim if dim >= 0 else dim + parts[0]._v.dim()) != 0:
                    return parts[0]._wrap(torch.cat([t._v for t in parts], dim=dim))
        # anything else sees ordinary tensors in original edge order
        conv = lambda a: a.materialize() if isinstance(a, EdgeTensor) else a
        res = func(*[conv(a) if not isinstance(a, (list, tuple)) else type(a)(conv(x) for x in a) for a in args],
                   **{k: conv(v) for k, v in kwargs.items()})
        out = kwargs.get("out")
        if isinstance(out, EdgeTensor):               # func wrote into the original-order copy: the sorted rows follow
            out._resync()
            return out
        name = getattr(func, "__name__", "")
        if name.endswith("_") and not name.endswith("__") and args and isinstance(args[0], EdgeTensor):   # torch.Tensor.mul_(et, ...) and friends
            args[0]._resync()
            return args[0]
        return res


def _known_prod(shape):
    p = 1
    for s in shape:
        if s == -1:
            return -1
        p *= int(s)
    return p


def _only_dtype(a, k):
    if k and set(k) - {"dtype"}:
        return False
    return all(isinstance(x, torch.dtype) for x in a) and all(isinstance(v, torch.dtype) for v in k.values())


def _dims(dim, nd):
    if dim is None:
        return None
    dims = dim if isinstance(dim, (tuple, list)) else (dim,)
    return [d if d >= 0 else d + nd for d in dims]


def materialize(t):
    """t as an ordinary original-order tensor (identity for anything that is not an EdgeTensor)."""
    return t.materialize() if isinstance(t, EdgeTensor) else t


def sorted_rows(t, graph):
    """The destination-sorted rows of t if it is an EdgeTensor of `graph`, else None."""
    return t.sorted_rows(graph) if isinstance(t, EdgeTensor) else None
