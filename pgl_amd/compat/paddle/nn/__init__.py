"""paddle.nn over torch.nn: Layer (= Module + create_parameter / sublayers), LayerList, Linear, Dropout, LeakyReLU,
functional, loss, initializer."""
import torch as _t
import torch.nn as _nn

from pgl_amd.nn.conv import _Linear as _TallLinear

from . import functional, initializer, loss                            # noqa: F401
from .loss import CrossEntropyLoss                                     # noqa: F401


class Layer(_nn.Module):
    def create_parameter(self, shape, attr=None, dtype="float32", is_bias=False, default_initializer=None):
        p = _nn.Parameter(_t.zeros([int(s) for s in shape]))
        init = default_initializer or (initializer.Constant(0.0) if is_bias else initializer.XavierUniform())
        init(p)
        return p

    def sublayers(self, include_self=False):
        mods = list(self.modules())
        return mods if include_self else mods[1:]

    def clear_gradients(self):
        self.zero_grad(set_to_none=True)


class LayerList(_nn.ModuleList, Layer):
    pass


class Linear(_TallLinear, Layer):
    """paddle.nn.Linear(in_features, out_features): Xavier-uniform weight, zero bias (Paddle's defaults).  The weight is
    stored torch-style [out, in]; Paddle's state dicts hold [in, out].  (The engine's Linear: same parameters and values as
    torch.nn.Linear, with the weight and bias gradients of inputs of millions of rows computed in split reductions -- the
    classifier head of examples/gcn/train.py at |V| = 2^20 spent 8.8 ms of a 19 ms training step in those two gradients.)"""

    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        _TallLinear.__init__(self, int(in_features), int(out_features), bias=bias_attr is not False)
        _nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            _nn.init.zeros_(self.bias)


class Dropout(_nn.Dropout, Layer):
    def __init__(self, p=0.5, axis=None, mode="upscale_in_train", name=None):
        _nn.Dropout.__init__(self, p=float(p))


class LeakyReLU(_nn.LeakyReLU, Layer):
    def __init__(self, negative_slope=0.01, name=None):
        _nn.LeakyReLU.__init__(self, negative_slope=float(negative_slope))


class ReLU(_nn.ReLU, Layer):
    def __init__(self, name=None):
        _nn.ReLU.__init__(self)
