"""paddle.nn.functional names used by pgl.nn.conv and the examples."""
import torch as _t
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
    label = label.long()
    if weight is None and not soft_label and input.dim() == 2 and axis in (-1, 1) and reduction in ("mean", "sum", "none"):
        # log-softmax, then ONE gathered element per row.  torch's nll_loss kernels take 1.8 ms forward + 1.5 ms backward for a
        # [2^20, 41] input (the classifier output of examples/gcn/train.py at |V| = 2^20); the gather and its scatter backward
        # are 0.1 ms each.  Same values: -log p[label], rows with label == ignore_index excluded from sum and count.
        logp = _F.log_softmax(input, dim=-1)
        keep = label != ignore_index
        picked = -logp.gather(-1, label.clamp(min=0).unsqueeze(-1)).squeeze(-1)
        picked = _t.where(keep, picked, picked.new_zeros(()))
        if reduction == "none":
            return picked
        return picked.sum() if reduction == "sum" else picked.sum() / keep.sum().clamp(min=1).to(picked.dtype)
    return _F.cross_entropy(input, label, weight=weight, ignore_index=ignore_index, reduction=reduction)
