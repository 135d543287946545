"""Minimal `paddle` namespace over PyTorch-ROCm -- the names the reference's gcn / gat / graphsage examples and pgl.nn.conv
use.  Tensors are torch tensors on the current MI355X; see pgl_amd/compat/__init__.py.

Differences from torch that the scripts rely on and that are therefore adapted here:
  * paddle.to_tensor puts data on the accelerator by default; numpy float64 stays float64, python floats become float32;
  * paddle.gather(x, index) accepts an index of shape [k] or [k, 1];
  * CrossEntropyLoss / accuracy take integer labels of shape [k, 1];
  * Tensor.numpy() works on device tensors that require grad (Paddle detaches and copies);
  * optimizer.clear_grad(), Layer.create_parameter, Layer.sublayers, LayerList.
"""
import contextlib as _contextlib
import functools as _functools

import numpy as _np
import torch as _t

from . import nn, optimizer, metric, distributed, device, static      # noqa: F401  (submodule imports the scripts use)

__version__ = "2.4.2+pgl_amd.compat"
Tensor = _t.Tensor
float16, float32, float64, bfloat16 = _t.float16, _t.float32, _t.float64, _t.bfloat16
int8, int16, int32, int64, uint8 = _t.int8, _t.int16, _t.int32, _t.int64, _t.uint8

_DT = {"float16": _t.float16, "bfloat16": _t.bfloat16, "float32": _t.float32, "float64": _t.float64, "int8": _t.int8,
       "int16": _t.int16, "int32": _t.int32, "int64": _t.int64, "uint8": _t.uint8, "bool": _t.bool}
_default_dtype = ["float32"]


def _dtype(d):
    if d is None or isinstance(d, _t.dtype):
        return d
    if isinstance(d, str):
        return _DT[d]
    return _DT[_np.dtype(d).name]


def _device():
    return _t.device("cuda", _t.cuda.current_device()) if _t.cuda.is_available() else _t.device("cpu")


# Paddle creates parameters and factory tensors on the current accelerator ("place") by default, and the reference's scripts
# never move a model: make torch's factory functions do the same while this name layer is in use.
if _t.cuda.is_available():
    _t.set_default_device(_device())


def get_default_dtype():
    return _default_dtype[0]


def set_default_dtype(d):
    name = d if isinstance(d, str) else str(_dtype(d)).replace("torch.", "")
    _default_dtype[0] = name
    _t.set_default_dtype(_DT[name])


def to_tensor(data, dtype=None, place=None, stop_gradient=True):
    if isinstance(data, _t.Tensor):
        t = data.to(_device())
    else:
        a = _np.asarray(data)
        if a.dtype == _np.float64 and not isinstance(data, _np.ndarray):
            a = a.astype(_default_dtype[0])               # python floats / lists of floats: the default dtype
        t = _t.as_tensor(a).to(_device())
    if dtype is not None:
        t = t.to(_dtype(dtype))
    if not stop_gradient and t.is_floating_point():
        t.requires_grad_(True)
    return t


def is_tensor(x):
    return isinstance(x, _t.Tensor)


def _index(index):
    index = index.reshape(-1) if index.dim() > 1 else index
    return index.long() if index.dtype != _t.int64 else index


def gather(x, index, axis=0, name=None):
    return _t.index_select(x, int(axis), _index(index))


def reshape(x, shape, name=None):
    return x.reshape([int(s) for s in shape])


def concat(x, axis=0, name=None):
    return _t.cat(list(x), dim=int(axis))


def sum(x, axis=None, dtype=None, keepdim=False, name=None):          # noqa: A001
    return x.sum(dtype=_dtype(dtype)) if axis is None else x.sum(dim=axis, keepdim=keepdim, dtype=_dtype(dtype))


def mean(x, axis=None, keepdim=False, name=None):
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=keepdim)


def zeros(shape, dtype=None):
    return _t.zeros([int(s) for s in shape], dtype=_dtype(dtype or _default_dtype[0]), device=_device())


def ones(shape, dtype=None):
    return _t.ones([int(s) for s in shape], dtype=_dtype(dtype or _default_dtype[0]), device=_device())


def arange(start=0, end=None, step=1, dtype=None):
    if end is None:
        start, end = 0, start
    return _t.arange(start, end, step, dtype=_dtype(dtype or "int64"), device=_device())


def randn(shape, dtype=None, name=None):
    return _t.randn([int(v) for v in shape], dtype=_dtype(dtype or _default_dtype[0]), device=_device())


def scatter(x, index, updates, overwrite=True, name=None):
    """paddle.scatter: rows `index` of a COPY of x replaced by (overwrite) or, after zeroing them, summed with `updates`."""
    out = x.clone()
    idx = _index(index)
    upd = updates.to(out.dtype)
    if overwrite:
        out[idx] = upd
    else:
        out[idx] = 0
        out.index_add_(0, idx, upd)
    return out


def seed(s):
    _t.manual_seed(int(s))
    return s


class no_grad(_contextlib.ContextDecorator):                          # noqa: N801 -- usable as `with` and as `@paddle.no_grad()`
    def __enter__(self):
        self._prev = _t.is_grad_enabled()
        _t.set_grad_enabled(False)

    def __exit__(self, *exc):
        _t.set_grad_enabled(self._prev)
        return False


class DataParallel(object):
    """paddle.DataParallel(model) (examples/graphsage/cpu_sample_version/train.py:116): a transparent wrapper on one GPU;
    with an initialised process group, torch's DistributedDataParallel (gradient all-reduce over RCCL)."""

    def __new__(cls, layers, *a, **k):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return _t.nn.parallel.DistributedDataParallel(layers)
        return layers


# ---- Tensor methods Paddle code calls on results (kept to what the examples and pgl.nn use) -----------------------------
_torch_numpy = _t.Tensor.numpy


def _numpy(self, *a, **k):
    return _torch_numpy(self.detach().cpu(), *a, **k)


_t.Tensor.numpy = _numpy

# Tensor.shape is a LIST in Paddle and the reference's code and tests compare it with lists (tests/test_graph.py:222): hand out a
# tuple that also equals the LIST of the same numbers (torch.Size cannot be subclassed; it is itself a tuple, and everything torch
# does with a shape -- sizes of factory calls, slicing, unpacking, numel() -- works on this one too).
class _Shape(tuple):
    def __eq__(self, other):
        return tuple.__eq__(self, tuple(other)) if isinstance(other, (list, tuple)) else NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = tuple.__hash__

    def numel(self):
        n = 1
        for v in self:
            n *= int(v)
        return n

    def __repr__(self):
        return "paddle.shape(%s)" % list(self)


_torch_shape = _t.Tensor.shape
_t.Tensor.shape = property(lambda self: _Shape(_torch_shape.__get__(self)))

# Tensor.transpose(perm) takes the whole permutation in Paddle (tests/test_pool.py:125); torch's takes two axes
_torch_transpose = _t.Tensor.transpose


def _transpose(self, *a, **k):
    if len(a) == 1 and isinstance(a[0], (list, tuple)):
        return self.permute(*[int(v) for v in a[0]])
    if "perm" in k:
        return self.permute(*[int(v) for v in k["perm"]])
    return _torch_transpose(self, *a, **k)


_t.Tensor.transpose = _transpose
_t.Tensor.astype = lambda self, d: self.to(_dtype(d))
_t.Tensor.clear_gradient = lambda self: setattr(self, "grad", None)
_t.Tensor.stop_gradient = property(lambda self: not self.requires_grad,
                                   lambda self, v: self.requires_grad_(not v) if (self.is_floating_point() and self.is_leaf) else None)
