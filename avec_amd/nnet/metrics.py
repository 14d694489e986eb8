"""Metrics (nnet/metrics.py): host-side text metric; jiwer is not a dependency -- word-level Levenshtein distance restated."""
import torch.nn as nn


def _edit_distance(r, h):
    d = list(range(len(h) + 1))
    for i in range(1, len(r) + 1):
        prev, d[0] = d[0], i
        for j in range(1, len(h) + 1):
            cur = d[j]
            d[j] = min(d[j] + 1, d[j - 1] + 1, prev + (r[i - 1] != h[j - 1]))
            prev = cur
    return d[len(h)]


class WordErrorRate(nn.Module):
    """100 * (S + D + I) / N over the batch (corpus-level, like jiwer.wer on lists)."""

    def __init__(self, name="wer"):
        super().__init__()
        self.name = name

    def forward(self, targets, outputs):
        errs = words = 0
        for t, o in zip(targets, outputs):
            tw = t.split() if isinstance(t, str) else list(t)
            ow = o.split() if isinstance(o, str) else list(o)
            errs += _edit_distance(tw, ow)
            words += len(tw)
        return 100.0 * errs / max(words, 1)


metric_dict = {"WordErrorRate": WordErrorRate}
