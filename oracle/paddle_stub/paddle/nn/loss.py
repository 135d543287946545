import torch
import torch.nn as tnn


class CrossEntropyLoss(tnn.Module):
    def __init__(self, weight=None, ignore_index=-100, reduction="mean", soft_label=False, axis=-1, **kw):
        super().__init__()
        self.reduction, self.ignore_index = reduction, ignore_index

    def forward(self, input, label):  # noqa: A002
        return torch.nn.functional.cross_entropy(input, label.reshape(-1).to(torch.int64), ignore_index=self.ignore_index,
                                                 reduction=self.reduction)
