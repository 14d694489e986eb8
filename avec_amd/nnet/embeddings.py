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


class GroupedRelativeSinusoidalPositionalEncoding(nn.Module):
    """nnet/embeddings.py:160-216: the sinusoid rows a grouped attention over G-frame tokens needs.  Full context: rows for the frame offsets
    p = (T + Th - 1 - G//2) ... -(T - 1 - G//2), i.e. Th + 2T - G rows (a multiple of G when T and Th are), regrouped G at a time by the attention.
    Generated on demand (no (2*max_len - G%2, D) buffer per layer)."""

    def __init__(self, max_len, dim_model, group_size=1, causal=False):
        super().__init__()
        assert not causal, "causal grouped relative positions are not on this path (the encoders pass causal=False, nnet/networks.py:330)"
        self.max_len, self.dim_model, self.group_size, self.causal = max_len, dim_model, group_size, causal

    def forward(self, batch_size=1, seq_len=None, hidden_len=0, device="cuda"):
        import torch
        G, D = self.group_size, self.dim_model
        T = self.max_len if seq_len is None else seq_len
        # reference table: positions max_len-1 .. (G%2) then 0 .. -(max_len-1) (for even G the position 0 occurs twice); slice
        # [max_len - T + G//2 - Th : max_len - G%2 + T - G//2)  ->  positions hi .. lo (with the doubled 0 for even G)
        hi, lo = T - 1 - G // 2 + hidden_len, -(T - G // 2 - 1)
        if G % 2:
            pos = torch.arange(hi, lo - 1, -1, dtype=torch.float32)
        else:
            pos = torch.cat([torch.arange(hi, -1, -1, dtype=torch.float32), torch.arange(0, lo - 1, -1, dtype=torch.float32)])
        pos = pos.unsqueeze(1)
        inv = 10000 ** (2 * torch.arange(0, D // 2, dtype=torch.float32).unsqueeze(0) / D)
        ang = pos / inv
        pe = torch.zeros(pos.shape[0], D)
        pe[:, 0::2] = ang.sin()
        pe[:, 1::2] = ang.cos()
        return pe.to(device).unsqueeze(0).expand(batch_size, -1, -1)
