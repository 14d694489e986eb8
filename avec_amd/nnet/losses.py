"""Losses (nnet/losses.py)."""
import torch
import torch.nn as nn

from .. import ops


class CTCLoss(nn.Module):
    """nnet/losses.py:292-334: log_softmax + CTC(blank, zero_infinity), summed over frames, mean over the batch ("mean"),
    or summed ("sum").  One HIP kernel computes the normalisers, alpha/beta recursions and d/dlogits."""

    def __init__(self, blank=0, reduction="mean", zero_infinity=False, assert_shorter=True):
        super().__init__()
        assert reduction in ("mean", "sum"), "hot path: 'mean' (sum over frames, mean over batch) / 'sum'"
        self.blank, self.reduction, self.zero_infinity, self.assert_shorter = blank, reduction, zero_infinity, assert_shorter

    def forward(self, targets, outputs):
        y, y_len = targets
        logits, logits_len = outputs
        if self.assert_shorter:
            assert (y_len.cpu() <= logits_len.cpu()).all(), "logits length shorter than label length"
        loss = ops.CTCLossFn.apply(logits, logits_len, y, y_len, self.blank, self.zero_infinity)
        return loss * logits.shape[0] if self.reduction == "sum" else loss


class SoftmaxCrossEntropy(nn.Module):
    """nnet/losses.py:258-290: nn.CrossEntropyLoss(ignore_index, reduction='none') followed by Reduction('mean' = mean over every element, ignored
    ones counting as 0 | 'sum').  logits (..., V), or (B, T, V) with transpose_logits=True (the reference transposes them to class-dim 1)."""

    def __init__(self, ignore_index=-1, transpose_logits=False, reduction="mean"):
        super().__init__()
        assert reduction in ("mean", "sum"), "hot path: 'mean' / 'sum'"
        self.ignore_index, self.transpose_logits, self.reduction = ignore_index, transpose_logits, reduction

    def forward(self, targets, outputs):
        logits = outputs
        assert logits.dim() == 2 or self.transpose_logits, "logits (M, V); (B, T, V) needs transpose_logits=True"
        V = logits.shape[-1]
        loss = ops.SoftmaxCEFn.apply(logits.reshape(-1, V), targets.reshape(-1), self.ignore_index)
        return loss * (logits.numel() // V) if self.reduction == "sum" else loss


loss_dict = {"CTC": CTCLoss, "SoftmaxCrossEntropy": SoftmaxCrossEntropy}
