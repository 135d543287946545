"""paddle.nn.functional names used by pgl.nn.conv and the examples."""
import torch.nn.functional as _F

from pgl_amd.nn.functional.loss import cross_entropy as _engine_cross_entropy

relu, elu, leaky_relu, softmax, log_softmax, sigmoid, tanh, gelu = (_F.relu, _F.elu, _F.leaky_relu, _F.softmax, _F.log_softmax,
                                                                    _F.sigmoid, _F.tanh, _F.gelu)


def normalize(x, p=2, axis=1, epsilon=1e-12, name=None):
    return _F.normalize(x, p=p, dim=axis, eps=epsilon)


def dropout(x, p=0.5, axis=None, training=True, mode="upscale_in_train", name=None):
    return _F.dropout(x, p=p, training=training)


def cross_entropy(input, label, weight=None, ignore_index=-100, reduction="mean", soft_label=False, axis=-1, name=None):   # noqa: A002
    if label.dim() == input.dim() and label.shape[-1] == 1:
        label = label.squeeze(-1)
    label = label.long()
    if weight is None and not soft_label and input.dim() == 2 and axis in (-1, 1) and reduction in ("mean", "sum", "none"):
        return _engine_cross_entropy(input, label, ignore_index, reduction)     # same values, without torch's slow nll_loss kernels
    return _F.cross_entropy(input, label, weight=weight, ignore_index=ignore_index, reduction=reduction)
