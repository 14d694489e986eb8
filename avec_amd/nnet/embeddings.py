"""Relative sinusoidal positional encoding (nnet/embeddings.py:101-158).  The reference keeps a (1, 2*max_len-1, D) table per attention layer
(24 copies, re-broadcast by DDP every step); here the 2T-1 rows needed are generated once per (T, D) and cached (ops.rel_pos_table)."""
import torch.nn as nn

from .. import ops


class RelativeSinusoidalPositionalEncoding(nn.Module):
    def __init__(self, max_len, dim_model, causal=False):
        super().__init__()
        self.max_len, self.dim_model, self.causal = max_len, dim_model, causal

    def forward(self, batch_size=1, seq_len=None, hidden_len=0, device="cuda"):
        """rows p = T-1 .. -(T-1) as (B, 2T-1, D) -- causal: only p = T-1 .. 0, the first T rows (nnet/embeddings.py:136-145); B copies are a view
        (the kernels never materialise them).  The attention kernels index the full table by i - j; under a causal mask (no key j > i visible) that reads
        exactly the causal rows, which is why ConformerInterCTC accepts causal=True only together with such a mask."""
        T = self.max_len if seq_len is None else seq_len
        assert hidden_len == 0, "attention state caches (hidden) are not part of this path"
        tab = ops.rel_pos_table(T, self.dim_model, device)
        if self.causal:
            tab = tab[:T]
        return tab.unsqueeze(0).expand(batch_size, -1, -1)
