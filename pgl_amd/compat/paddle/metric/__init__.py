"""paddle.metric.accuracy(input, label, k=1): top-k accuracy as a 0-d tensor; labels [n] or [n, 1]."""
import torch as _t


def accuracy(input, label, k=1, correct=None, total=None, name=None):   # noqa: A002
    label = label.reshape(-1, 1).long()
    topk = input.topk(int(k), dim=-1).indices
    return (topk == label).any(dim=-1).to(_t.float32).mean()
