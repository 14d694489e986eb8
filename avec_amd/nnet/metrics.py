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


_CONTRACTIONS = (("won't", "will not"), ("can't", "can not"), ("let's", "let us"), ("n't", " not"), ("'re", " are"), ("'s", " is"), ("'d", " would"),
                 ("'ll", " will"), ("'t", " not"), ("'ve", " have"), ("'m", " am"))


def standardize(text):
    """What jiwer.wer(..., standardize=True) applies before counting (nnet/metrics.py:110; jiwer is an un-vendored dependency -- its documented default
    transform restated: lower case, common English contractions expanded, Kaldi non-words `[..]` / `<..>` removed, white space collapsed).  "parity unpinned"."""
    import re
    text = text.lower()
    for a, b in _CONTRACTIONS:
        text = text.replace(a, b)
    text = re.sub(r"[<\[][^>\]]*[>\]]", "", text)
    return " ".join(text.split())


class WordErrorRate(nn.Module):
    """100 * (S + D + I) / N over the given sentences (corpus-level, like jiwer.wer on lists) after jiwer's `standardize` normalisation."""

    def __init__(self, name="wer"):
        super().__init__()
        self.name = name

    def forward(self, targets, outputs):
        errs = words = 0
        for t, o in zip(targets, outputs):
            tw = standardize(t).split() if isinstance(t, str) else list(t)
            ow = standardize(o).split() if isinstance(o, str) else list(o)
            errs += _edit_distance(tw, ow)
            words += len(tw)
        return 100.0 * errs / max(words, 1)


class CategoricalAccuracy(nn.Module):
    """nnet/metrics.py:40-69: 100 * (# argmax(y_pred) == y_true over positions whose target is not ignore_index) / (# such positions)."""

    def __init__(self, ignore_index=-1, dim_argmax=-1, name="acc"):
        super().__init__()
        self.name, self.dim_argmax, self.ignore_index = name, dim_argmax, ignore_index

    def forward(self, y_true, y_pred):
        import torch
        from .. import ops
        if self.dim_argmax is not None:
            assert self.dim_argmax in (-1, y_pred.dim() - 1), "argmax over the last axis"
            y_pred = ops.argmax_rows(y_pred.float()) if y_pred.is_cuda else y_pred.argmax(dim=-1)
        y_true = y_true.to(y_pred.device)
        keep = y_true != self.ignore_index
        n = int(keep.sum())
        return 100.0 * float(((y_pred.reshape(y_true.shape) == y_true) & keep).sum()) / max(n, 1)


metric_dict = {"WordErrorRate": WordErrorRate, "CategoricalAccuracy": CategoricalAccuracy}
