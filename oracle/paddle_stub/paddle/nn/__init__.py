"""paddle.nn subset over torch.nn (CPU).  Weight layout follows Paddle: Linear.weight is [in, out]."""
import math

import torch
import torch.nn as tnn

from . import functional, initializer  # noqa: F401
from . import loss  # noqa: F401
from .loss import CrossEntropyLoss  # noqa: F401


class Layer(tnn.Module):
    def __init__(self, name_scope=None, dtype="float32"):
        super().__init__()

    def create_parameter(self, shape, attr=None, dtype=None, is_bias=False, default_initializer=None):
        import paddle
        if attr is False:
            return None
        p = tnn.Parameter(torch.zeros([int(s) for s in shape], dtype=paddle._dt(dtype or paddle.get_default_dtype())))
        init = default_initializer or (getattr(attr, "initializer", None) if attr is not None else None)
        if init is not None:
            init(p)
        elif not is_bias:
            initializer.XavierUniform()(p)
        return p

    def add_sublayer(self, name, layer):
        self.add_module(name, layer)
        return layer

    def sublayers(self, include_self=False):
        return [m for m in self.modules() if include_self or m is not self]

    def set_state_dict(self, sd, *a, **k):
        return self.load_state_dict(sd)

    set_dict = set_state_dict

    def clear_gradients(self):
        for p in self.parameters():
            p.grad = None


class LayerList(tnn.ModuleList):
    pass


class Sequential(tnn.Sequential):
    pass


class Linear(Layer):
    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        self.weight = self.create_parameter([in_features, out_features], attr=weight_attr)
        self.bias = None if bias_attr is False else self.create_parameter([out_features], attr=bias_attr, is_bias=True)

    def forward(self, x):
        y = torch.matmul(x, self.weight)
        return y if self.bias is None else y + self.bias


class Embedding(Layer):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, sparse=False, weight_attr=None, name=None):
        super().__init__()
        self.weight = self.create_parameter([num_embeddings, embedding_dim], attr=weight_attr)

    def forward(self, ids):
        return self.weight[ids.to(torch.int64)]


class Dropout(Layer):
    def __init__(self, p=0.5, axis=None, mode="upscale_in_train", name=None):
        super().__init__()
        self.p = p

    def forward(self, x):
        return torch.nn.functional.dropout(x, self.p, self.training)


class LeakyReLU(Layer):
    def __init__(self, negative_slope=0.01, name=None):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):
        return torch.nn.functional.leaky_relu(x, self.negative_slope)


class ReLU(Layer):
    def forward(self, x):
        return torch.relu(x)


class Tanh(Layer):
    def forward(self, x):
        return torch.tanh(x)


class Sigmoid(Layer):
    def forward(self, x):
        return torch.sigmoid(x)


class BatchNorm1D(tnn.BatchNorm1d):
    def __init__(self, num_features, momentum=0.9, epsilon=1e-5, **kw):
        super().__init__(num_features, eps=epsilon, momentum=1 - momentum)


class LayerNorm(tnn.LayerNorm):
    def __init__(self, normalized_shape, epsilon=1e-5, **kw):
        super().__init__(normalized_shape, eps=epsilon)


class LSTM(Layer):
    def __init__(self, input_size, hidden_size, num_layers=1, direction="forward", time_major=False, dropout=0.0, **kw):
        super().__init__()
        self.lstm = tnn.LSTM(input_size, hidden_size, num_layers, batch_first=not time_major,
                             bidirectional=direction != "forward", dropout=dropout)

    def forward(self, x, initial_states=None, sequence_length=None):
        return self.lstm(x, initial_states)
