import math

import torch


class _Init:
    def __call__(self, p, block=None):
        raise NotImplementedError


def _fans(p):
    shape = p.shape
    if len(shape) < 2:
        return int(shape[0]), int(shape[0])
    rf = 1
    for s in shape[2:]:
        rf *= int(s)
    return int(shape[0]) * rf, int(shape[1]) * rf        # paddle: weight is [in, out]


class XavierUniform(_Init):
    def __init__(self, fan_in=None, fan_out=None, name=None):
        self.fi, self.fo = fan_in, fan_out

    def __call__(self, p, block=None):
        fi, fo = _fans(p)
        fi, fo = self.fi or fi, self.fo or fo
        lim = math.sqrt(6.0 / (fi + fo))
        with torch.no_grad():
            p.uniform_(-lim, lim)


class XavierNormal(_Init):
    def __call__(self, p, block=None):
        fi, fo = _fans(p)
        with torch.no_grad():
            p.normal_(0, math.sqrt(2.0 / (fi + fo)))


class KaimingUniform(_Init):
    def __init__(self, fan_in=None, **kw):
        self.fi = fan_in

    def __call__(self, p, block=None):
        fi, _ = _fans(p)
        lim = math.sqrt(6.0 / (self.fi or fi))
        with torch.no_grad():
            p.uniform_(-lim, lim)


class Constant(_Init):
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, p, block=None):
        with torch.no_grad():
            p.fill_(self.value)


class Uniform(_Init):
    def __init__(self, low=-1.0, high=1.0, name=None):
        self.low, self.high = low, high

    def __call__(self, p, block=None):
        with torch.no_grad():
            p.uniform_(self.low, self.high)


class Normal(_Init):
    def __init__(self, mean=0.0, std=1.0, name=None):
        self.mean, self.std = mean, std

    def __call__(self, p, block=None):
        with torch.no_grad():
            p.normal_(self.mean, self.std)


class Assign(_Init):
    def __init__(self, value, name=None):
        self.value = value

    def __call__(self, p, block=None):
        with torch.no_grad():
            p.copy_(torch.as_tensor(self.value, dtype=p.dtype).reshape(p.shape))
