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


loss_dict = {"CTC": CTCLoss}
