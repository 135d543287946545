"""Classifier-head pieces for node-level training at graph scale (engine extensions; the reference's examples reach them through
paddle.nn.functional.cross_entropy / paddle.nn.Linear, which pgl_amd/compat/paddle maps here)."""
import torch
import torch.nn.functional as F


def cross_entropy(input, label, ignore_index=-100, reduction="mean"):   # noqa: A002
    """-log softmax(input)[label] over rows of a [N, C] tensor, integer labels [N]; rows whose label is ignore_index count neither in
    the sum nor in the mean's denominator -- the values of torch.nn.functional.cross_entropy.  Computed as log-softmax + ONE gathered
    element per row: torch's nll_loss kernels take 1.8 ms forward + 1.5 ms backward on a [2^20, 41] input (the classifier output of
    examples/gcn/train.py at |V| = 2^20), the gather and its scatter backward 0.1 ms each."""
    label = label.long()
    logp = F.log_softmax(input, dim=-1)
    keep = label != ignore_index
    picked = -logp.gather(-1, torch.where(keep, label, label.new_zeros(())).unsqueeze(-1)).squeeze(-1)   # only ignored rows are redirected: any other out-of-range label still fails in the gather
    picked = torch.where(keep, picked, picked.new_zeros(()))
    if reduction == "none":
        return picked
    if reduction == "sum":
        return picked.sum()
    if reduction == "mean":
        return picked.sum() / keep.sum().clamp(min=1).to(picked.dtype)
    raise ValueError("cross_entropy: reduction must be 'mean', 'sum' or 'none'")
