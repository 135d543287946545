"""Minimal `paddle` stand-in, torch-CPU backed.  TEST INFRASTRUCTURE ONLY (lives under oracle/).

Purpose: let the REFERENCE's own Python package (/root/reference/pgl, imported read-only, never copied) run in the
build container, where PaddlePaddle is not installed, so that
  * the reference's own unit tests for the message-passing path can be executed against the CPU oracle
    (oracle/run_reference_tests.py), and
  * golden vectors for the layer glue (GCNConv / GATConv / GraphSageConv, send/recv with UDFs, edge_softmax, ...)
    can be produced BY THE REFERENCE'S CODE (tests/golden/make_golden_layers.py).
Dense tensor ops map to torch on CPU; the graph primitives the reference delegates to PaddlePaddle
(paddle.geometric.*) are answered by the oracle's restatement (oracle/ref_ops.py).  Nothing under pgl_amd/ imports this.
"""
import builtins as _b

import numpy as _np
import torch as _t

__version__ = "2.4.2"
Tensor = _t.Tensor

float16, float32, float64 = _t.float16, _t.float32, _t.float64
int8, int16, int32, int64, uint8 = _t.int8, _t.int16, _t.int32, _t.int64, _t.uint8
bool = _t.bool  # noqa: A001

_DT = {"float16": _t.float16, "float32": _t.float32, "float64": _t.float64, "int32": _t.int32, "int64": _t.int64,
       "int8": _t.int8, "int16": _t.int16, "uint8": _t.uint8, "bool": _t.bool, "float": _t.float32, "int": _t.int64}
_default_dtype = ["float32"]


def _dt(d):
    if d is None:
        return None
    if isinstance(d, _t.dtype):
        return d
    if isinstance(d, str):
        return _DT[d]
    if isinstance(d, type) and issubclass(d, _np.generic) or isinstance(d, _np.dtype):
        return _DT[_np.dtype(d).name]
    raise TypeError("dtype %r" % (d,))


def _shape(s):
    if isinstance(s, _t.Tensor):
        return [int(v) for v in s.reshape(-1).tolist()]
    if isinstance(s, (int, _np.integer)):
        return [int(s)]
    return [int(v) if not isinstance(v, _t.Tensor) else int(v.item()) for v in s]


# --- Tensor method aliases paddle code relies on ---------------------------------------------------------
def _astype(self, d):
    return self.to(_dt(d))


_t.Tensor.astype = _astype
_t.Tensor.cast = _astype
_orig_numpy = _t.Tensor.numpy
_t.Tensor.numpy = lambda self, *a, **k: _orig_numpy(self.detach().cpu(), *a, **k)
_t.Tensor.clear_gradient = lambda self: setattr(self, "grad", None)
_t.Tensor.stop_gradient = property(lambda self: not self.requires_grad,
                                   lambda self, v: self.requires_grad_(not v) if self.is_floating_point() and self.is_leaf else None)
_t.Tensor.place = property(lambda self: CPUPlace())
_t.Tensor.pin_memory = lambda self: self
_t.Tensor.cuda = lambda self, *a, **k: self
_t.Tensor.cpu = lambda self: self
_t.Tensor._is_initialized = lambda self: True
_t.Tensor.shape = property(lambda self: list(self.size()))      # paddle: Tensor.shape is a python list

_orig_transpose = _t.Tensor.transpose


def _transpose(self, *a, **k):            # paddle: x.transpose(perm); torch: x.transpose(d0, d1)
    if len(a) == 1 and isinstance(a[0], (list, tuple)):
        return self.permute(*a[0])
    if "perm" in k:
        return self.permute(*k["perm"])
    return _orig_transpose(self, *a, **k)


_t.Tensor.transpose = _transpose
_orig_getitem = _t.Tensor.__getitem__


def _getitem(self, idx):                  # paddle accepts int32 index tensors
    if isinstance(idx, _t.Tensor) and idx.dtype in (_t.int32, _t.int16, _t.int8):
        idx = idx.to(_t.int64)
    elif isinstance(idx, _t.Tensor) and idx.is_floating_point():
        # pgl/math.py:276-296 adds a float32 offset to an int64 argsort result and indexes with the sum: Paddle 2.x keeps the
        # LEFT operand's dtype in mixed elementwise arithmetic (int64 there), torch promotes to float.  Integral values only.
        assert (idx == idx.round()).all().item(), "float index tensor with non-integral values"
        idx = idx.to(_t.int64)
    return _orig_getitem(self, idx)


_t.Tensor.__getitem__ = _getitem
_orig_index_select = _t.Tensor.index_select


def _index_select_method(self, *a, **k):   # paddle: x.index_select(index, axis=0); torch: x.index_select(dim, index)
    if (a and isinstance(a[0], _t.Tensor)) or "axis" in k:
        index = a[0] if a else k["index"]
        axis = a[1] if len(a) > 1 else k.get("axis", 0)
        return _orig_index_select(self, axis, _idx(index))
    return _orig_index_select(self, *a, **k)


_t.Tensor.index_select = _index_select_method
_orig_split = _t.Tensor.split


def split(x, num_or_sections, axis=0, name=None):      # paddle: an int is the NUMBER of sections
    if isinstance(num_or_sections, int):
        return list(_t.chunk(x, num_or_sections, dim=int(axis)))
    return list(_orig_split(x, [int(v) for v in num_or_sections], int(axis)))


_t.Tensor.split = split
# paddle.Tensor defines no in-place arithmetic dunders: `a += b` rebinds `a` to a NEW tensor (aliases keep the old
# value).  torch's own Python code (optimizers: `step_t += 1`) relies on the in-place meaning, so the Paddle meaning is
# applied only when the statement is NOT executed from inside the torch package.
import sys as _sys


def _paddle_inplace(orig, pure):
    def op(self, other):
        if _sys._getframe(1).f_globals.get("__name__", "").startswith("torch"):
            return orig(self, other)
        return pure(self, other)
    return op


_t.Tensor.__iadd__ = _paddle_inplace(_t.Tensor.__iadd__, lambda a, b: a + b)
_t.Tensor.__isub__ = _paddle_inplace(_t.Tensor.__isub__, lambda a, b: a - b)
_t.Tensor.__imul__ = _paddle_inplace(_t.Tensor.__imul__, lambda a, b: a * b)
_t.Tensor.__itruediv__ = _paddle_inplace(_t.Tensor.__itruediv__, lambda a, b: a / b)


class CPUPlace:
    def __repr__(self):
        return "Place(cpu)"


class CUDAPlace(CPUPlace):
    def __init__(self, i=0):
        self.i = i


class CUDAPinnedPlace(CPUPlace):
    pass


class ParamAttr:
    def __init__(self, name=None, initializer=None, learning_rate=1.0, regularizer=None, trainable=True, **kw):
        self.name, self.initializer, self.trainable = name, initializer, trainable


def get_default_dtype():
    return _default_dtype[0]


def set_default_dtype(d):
    _default_dtype[0] = d if isinstance(d, str) else str(_dt(d)).replace("torch.", "")


def is_tensor(x):
    return isinstance(x, _t.Tensor)


def to_tensor(data, dtype=None, place=None, stop_gradient=True):
    if isinstance(data, _t.Tensor):
        out = data.clone() if dtype is None else data.to(_dt(dtype))
    else:
        arr = _np.asarray(data)
        if dtype is None and arr.dtype == _np.float64 and not isinstance(data, _np.ndarray):
            arr = arr.astype(get_default_dtype())          # python floats -> default dtype, as paddle does
        out = _t.tensor(_np.ascontiguousarray(arr))          # a copy, as paddle.to_tensor makes
        if dtype is not None:
            out = out.to(_dt(dtype))
    return out


tensor = to_tensor


def _idx(i):
    return i.to(_t.int64) if isinstance(i, _t.Tensor) else _t.as_tensor(_np.asarray(i), dtype=_t.int64)


def gather(x, index, axis=None, name=None):
    index = _idx(index).reshape(-1)
    return x.index_select(0 if axis is None else int(axis), index)


def scatter(x, index, updates, overwrite=True, name=None):
    index = _idx(index).reshape(-1)
    out = x.clone()
    if overwrite:
        out[index] = updates.to(out.dtype)
    else:
        out[index] = 0
        out.index_add_(0, index, updates.to(out.dtype))
    return out


def scatter_nd(index, updates, shape, name=None):
    out = _t.zeros(_shape(shape), dtype=updates.dtype)
    index = _idx(index)
    out.index_put_(tuple(index[..., k] for k in range(index.shape[-1])), updates, accumulate=True)
    return out


def reshape(x, shape, name=None):
    return x.reshape(_shape(shape))


def concat(x, axis=0, name=None):
    return _t.cat(list(x), dim=int(axis))


def stack(x, axis=0, name=None):
    return _t.stack(list(x), dim=int(axis))


def zeros(shape, dtype=None, name=None):
    return _t.zeros(_shape(shape), dtype=_dt(dtype or get_default_dtype()))


def ones(shape, dtype=None, name=None):
    return _t.ones(_shape(shape), dtype=_dt(dtype or get_default_dtype()))


def full(shape, fill_value, dtype=None, name=None):
    if isinstance(fill_value, _t.Tensor):
        fill_value = fill_value.reshape(-1)[0].item()
    return _t.full(_shape(shape), fill_value, dtype=_dt(dtype or get_default_dtype()))


def empty(shape, dtype=None, name=None):
    return _t.zeros(_shape(shape), dtype=_dt(dtype or get_default_dtype()))


def zeros_like(x, dtype=None, name=None):
    return _t.zeros_like(x, dtype=_dt(dtype))


def ones_like(x, dtype=None, name=None):
    return _t.ones_like(x, dtype=_dt(dtype))


def arange(start=0, end=None, step=1, dtype=None, name=None):
    def v(a):
        return a.item() if isinstance(a, _t.Tensor) else a
    if end is None:
        start, end = 0, start
    return _t.arange(v(start), v(end), v(step), dtype=_dt(dtype or "int64"))


def shape(x):
    return _t.as_tensor(list(x.shape), dtype=_t.int32)


def cast(x, dtype):
    return x.to(_dt(dtype))


def _red(fn):
    def f(x, axis=None, keepdim=False, name=None, dtype=None):
        if axis is None:
            r = fn(x)
            return r.reshape([1] * x.dim()) if keepdim else r
        r = fn(x, dim=axis if isinstance(axis, int) else tuple(axis), keepdim=keepdim)
        return r[0] if isinstance(r, tuple) else r
    return f


sum = _red(_t.sum)  # noqa: A001
mean = _red(_t.mean)
max = _red(lambda x, **k: _t.amax(x, **k) if k else _t.max(x))  # noqa: A001
min = _red(lambda x, **k: _t.amin(x, **k) if k else _t.min(x))  # noqa: A001


def cumsum(x, axis=None, dtype=None, name=None):
    r = _t.cumsum(x.reshape(-1) if axis is None else x, dim=0 if axis is None else axis)
    return r if dtype is None else r.to(_dt(dtype))


def argsort(x, axis=-1, descending=False, name=None):
    return _t.argsort(x, dim=axis, descending=descending, stable=True)


def unique(x, return_index=False, return_inverse=False, return_counts=False, axis=None, dtype="int64", name=None):
    arr = x.numpy()
    res = _np.unique(arr, return_index=return_index, return_inverse=return_inverse, return_counts=return_counts, axis=axis)
    if not isinstance(res, tuple):
        return _t.as_tensor(res)
    return tuple(_t.as_tensor(_np.ascontiguousarray(r)) if i == 0 else _t.as_tensor(_np.ascontiguousarray(r)).to(_dt(dtype))
                 for i, r in enumerate(res))


def masked_select(x, mask, name=None):
    return _t.masked_select(x, mask)


def transpose(x, perm, name=None):
    return x.permute(*perm)


def matmul(x, y, transpose_x=False, transpose_y=False, name=None):
    if transpose_x:
        x = x.transpose(-1, -2)
    if transpose_y:
        y = y.transpose(-1, -2)
    return _t.matmul(x, y)


def clip(x, min=None, max=None, name=None):  # noqa: A002
    return _t.clamp(x, min=min, max=max)


def pow(x, y, name=None):  # noqa: A001
    return _t.pow(x, y)


def unsqueeze(x, axis, name=None):
    return x.unsqueeze(axis if isinstance(axis, int) else axis[0])


def squeeze(x, axis=None, name=None):
    return x.squeeze() if axis is None else x.squeeze(axis if isinstance(axis, int) else axis[0])


def multiply(x, y, name=None):
    return x * y


def add(x, y, name=None):
    return x + y


def index_select(x, index, axis=0, name=None):
    return x.index_select(axis, _idx(index))


def where(cond, x=None, y=None, name=None):
    return _t.where(cond) if x is None else _t.where(cond, x, y)


def randperm(n, dtype="int64", name=None):
    return _t.randperm(int(n)).to(_dt(dtype))


def randn(shape, dtype=None, name=None):
    return _t.randn(_shape(shape), dtype=_dt(dtype or get_default_dtype()))


def rand(shape, dtype=None, name=None):
    return _t.rand(_shape(shape), dtype=_dt(dtype or get_default_dtype()))


def seed(s):
    _t.manual_seed(int(s))
    _np.random.seed(int(s) % (2 ** 32))


exp, sqrt, tanh, log, abs, sigmoid = _t.exp, _t.sqrt, _t.tanh, _t.log, _t.abs, _t.sigmoid  # noqa: A001
no_grad = _t.no_grad
equal = lambda x, y: x == y  # noqa: E731
equal_all = lambda x, y: _t.as_tensor(_t.equal(x, y))  # noqa: E731


def set_device(d):
    return CPUPlace()


def get_device():
    return "cpu"


def disable_static(place=None):
    return None


def enable_static():
    raise RuntimeError("static graph mode is not available in the oracle's paddle stand-in")


def in_dynamic_mode():
    return True


def set_flags(flags):
    return None


def is_compiled_with_cuda():
    return False


def save(obj, path):
    _t.save(obj, path)


def load(path):
    return _t.load(path)


class DataParallel:
    def __new__(cls, layer, *a, **k):
        return layer


from . import nn, distributed, geometric, framework, device, static, optimizer, metric, io, incubate, common_ops_import  # noqa: E402,F401
from . import _C_ops, _legacy_C_ops  # noqa: E402,F401
