"""paddle.nn.initializer: callables applied in place to a parameter."""
import torch.nn as _nn


class Constant(object):
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, p, block=None):
        _nn.init.constant_(p, self.value)
        return p


class XavierUniform(object):
    def __init__(self, fan_in=None, fan_out=None, name=None):
        pass

    def __call__(self, p, block=None):
        if p.dim() >= 2:
            _nn.init.xavier_uniform_(p)
        else:
            _nn.init.zeros_(p)
        return p


class Normal(object):
    def __init__(self, mean=0.0, std=1.0, name=None):
        self.mean, self.std = mean, std

    def __call__(self, p, block=None):
        _nn.init.normal_(p, self.mean, self.std)
        return p
