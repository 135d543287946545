"""paddle.nn.functional names used by pgl.nn.conv and the examples."""
import torch.nn.functional as _F

relu, elu, leaky_relu, softmax, log_softmax, sigmoid, tanh, gelu = (_F.relu, _F.elu, _F.leaky_relu, _F.softmax, _F.log_softmax,
                                                                    _F.sigmoid, _F.tanh, _F.gelu)


def normalize(x, p=2, axis=1, epsilon=1e-12, name=None):
    return _F.normalize(x, p=p, dim=axis, eps=epsilon)


def dropout(x, p=0.5, axis=None, training=True, mode="upscale_in_train", name=None):
    return _F.dropout(x, p=p, training=training)


def cross_entropy(input, label, weight=None, ignore_index=-100, reduction="mean", soft_label=False, axis=-1, name=None):   # noqa: A002
    if label.dim() == input.dim() and label.shape[-1] == 1:
        label = label.squeeze(-1)
    return _F.cross_entropy(input, label.long(), weight=weight, ignore_index=ignore_index, reduction=reduction)
