import torch
import torch.nn.functional as TF


def relu(x, name=None):
    return torch.relu(x)


def elu(x, alpha=1.0, name=None):
    return TF.elu(x, alpha)


def leaky_relu(x, negative_slope=0.01, name=None):
    return TF.leaky_relu(x, negative_slope)


def sigmoid(x, name=None):
    return torch.sigmoid(x)


def tanh(x, name=None):
    return torch.tanh(x)


def softmax(x, axis=-1, dtype=None, name=None):
    return TF.softmax(x, dim=axis)


def log_softmax(x, axis=-1, dtype=None, name=None):
    return TF.log_softmax(x, dim=axis)


def dropout(x, p=0.5, axis=None, training=True, mode="upscale_in_train", name=None):
    return TF.dropout(x, p, training)


def normalize(x, p=2, axis=1, epsilon=1e-12, name=None):
    return TF.normalize(x, p=p, dim=axis, eps=epsilon)


def cross_entropy(input, label, weight=None, ignore_index=-100, reduction="mean", soft_label=False, axis=-1, **kw):  # noqa: A002
    return TF.cross_entropy(input, label.reshape(-1).to(torch.int64), weight=weight, ignore_index=ignore_index, reduction=reduction)


def softmax_with_cross_entropy(logits, label, **kw):
    return TF.cross_entropy(logits, label.reshape(-1).to(torch.int64), reduction="none").unsqueeze(-1)


def one_hot(x, num_classes, name=None):
    return TF.one_hot(x.to(torch.int64), num_classes).to(torch.float32)


def embedding(x, weight, **kw):
    return weight[x.to(torch.int64)]
