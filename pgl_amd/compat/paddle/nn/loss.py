"""paddle.nn.loss.CrossEntropyLoss: integer labels of shape [k] or [k, 1]."""
import torch.nn as _nn

from . import functional as _F


class CrossEntropyLoss(_nn.Module):
    def __init__(self, weight=None, ignore_index=-100, reduction="mean", soft_label=False, axis=-1, name=None):
        super(CrossEntropyLoss, self).__init__()
        self.weight, self.ignore_index, self.reduction = weight, ignore_index, reduction

    def forward(self, input, label):                                   # noqa: A002
        return _F.cross_entropy(input, label, self.weight, self.ignore_index, self.reduction)
