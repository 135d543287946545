import torch


def accuracy(input, label, k=1, correct=None, total=None, name=None):  # noqa: A002
    pred = input.argmax(-1)
    return (pred == label.reshape(-1).to(pred.dtype)).to(torch.float32).mean()
