"""Batch collation (nnet/collate_fn.py:31-176): pick tuple axes of each sample, zero-pad variable-length tensors to the batch maximum."""
import torch
import torch.nn as nn


class CollateFn(nn.Module):
    def __init__(self, inputs_params=[{"axis": 0}], targets_params=[{"axis": 1}]):
        super().__init__()
        self.inputs_params, self.targets_params = inputs_params, targets_params

    def forward(self, samples):
        return {"inputs": self.collate(samples, self.inputs_params), "targets": self.collate(samples, self.targets_params)}

    def collate(self, samples, params):
        single = isinstance(params, dict)
        out = []
        for p in ([params] if single else params):
            items = [s[p["axis"]] for s in samples]
            items = [torch.as_tensor(i) for i in items]
            if p.get("padding", False):
                out.append(torch.nn.utils.rnn.pad_sequence(items, batch_first=True, padding_value=p.get("padding_value", 0)))
            else:
                out.append(torch.stack(items))
        if single:
            return out[0]
        return tuple(out) if isinstance(params, tuple) else out
