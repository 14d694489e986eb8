"""Relative sinusoidal positional encoding (nnet/embeddings.py:101-158).  The reference keeps a (1, 2*max_len-1, D) table per attention layer
(24 copies, re-broadcast by DDP every step); here the 2T-1 rows needed are generated once per (T, D) and cached (ops.rel_pos_table)."""
import torch.nn as nn

from .. import ops


class RelativeSinusoidalPositionalEncoding(nn.Module):
    def __init__(self, max_len, dim_model, causal=False):
        super().__init__()
        assert not causal, "causal (streaming) relative positions are not on the hot path yet (SURVEY 8f rank 4)"
        self.max_len, self.dim_model, self.causal = max_len, dim_model, causal

    def forward(self, batch_size=1, seq_len=None, hidden_len=0, device="cuda"):
        """rows p = T-1 .. -(T-1) as (B, 2T-1, D); B copies are a view (the kernels never materialise them)."""
        T = self.max_len if seq_len is None else seq_len
        assert hidden_len == 0
        return ops.rel_pos_table(T, self.dim_model, device).unsqueeze(0).expand(batch_size, -1, -1)
