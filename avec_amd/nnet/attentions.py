"""Attention layers and masks (nnet/attentions.py): constructors, registries and state_dict keys of the reference; the computation
is one fused HIP sequence (ops.AttentionModuleFn)."""
import torch
import torch.nn as nn

from .. import ops
from .. import runtime as rt
from . import embeddings, layers


class MultiHeadAttention(nn.Module):
    """Parameter container with the reference layout: query/key/value/output Linear layers (nnet/attentions.py:27-171)."""

    def __init__(self, dim_model, num_heads, attn_drop_rate, weight_init="scaled_uniform", bias_init="zeros", output_proj=True, dim_kv=None):
        super().__init__()
        assert dim_kv in (None, dim_model) and output_proj, "cross-attention / projection-free variants are not on the hot path"
        assert attn_drop_rate == 0, "attention-probability dropout is 0 in every shipped config (nnet/networks.py:326,451,527)"
        self.num_heads, self.dim_model, self.dim_head = num_heads, dim_model, dim_model // num_heads
        self.output_proj, self.dim_kv = output_proj, dim_model
        self.dropout = nn.Identity()
        self.init_layers(weight_init, bias_init)

    def init_layers(self, weight_init, bias_init):
        D = self.dim_model
        self.query_layer = layers.Linear(D, D, weight_init=weight_init, bias_init=bias_init)
        self.key_layer = layers.Linear(D, D, weight_init=weight_init, bias_init=bias_init)
        self.value_layer = layers.Linear(D, D, weight_init=weight_init, bias_init=bias_init)
        rt.fuse_linears(self, (self.query_layer.weight, self.key_layer.weight, self.value_layer.weight),
                        (self.query_layer.bias, self.key_layer.bias, self.value_layer.bias))      # one Q|K|V GEMM per pass (runtime.FusedLinears)
        self.output_layer = layers.Linear(D, D, weight_init=weight_init, bias_init=bias_init)

    patch_size = 1

    def _params(self):
        return (self.query_layer.weight, self.query_layer.bias, self.key_layer.weight, self.key_layer.bias,
                self.value_layer.weight, self.value_layer.bias, self.output_layer.weight, self.output_layer.bias,
                self.pos_layer.weight, self.pos_layer.bias)

    def fused(self, x, ln, mask, lengths, drop_p, sid, residual):
        """y = [x +] Drop(attention(LN(x)))  -- called by AttentionModule; `lengths` (B,) is the fast path for key-padding masks."""
        if lengths is not None:
            lengths = lengths.to(device=x.device, dtype=torch.int64).contiguous()
            mask = None
        lw, lb, eps = (ln.weight, ln.bias, ln.eps) if ln is not None else (None, None, 0.0)
        return ops.AttentionModuleFn.apply(x, lengths, mask, lw, lb, *self._params(), self.num_heads, self.patch_size, eps, drop_p, sid, residual)

    def forward(self, x, mask=None, return_att_w=False):
        return self.forwardQKV(x, x, x, mask, return_att_w)

    def forwardQKV(self, Q, K, V, mask=None, return_att_w=False):
        assert Q is K and K is V, "only self-attention is on the hot path"
        assert not return_att_w, "attention weights are never materialised by the fused kernel"
        return self.fused(Q, None, mask, None, 0.0, 0, False)


class RelPos1dMultiHeadAttention(MultiHeadAttention):
    """nnet/attentions.py:215-323"""

    def __init__(self, dim_model, num_heads, num_pos_embeddings, attn_drop_rate, weight_init="scaled_uniform", bias_init="zeros", output_proj=True, causal=False):
        super().__init__(dim_model, num_heads, attn_drop_rate, weight_init=weight_init, bias_init=bias_init, output_proj=output_proj)
        self.causal = causal
        self.rel_pos_enc = embeddings.RelativeSinusoidalPositionalEncoding(num_pos_embeddings, dim_model, causal)
        self.pos_layer = layers.Linear(dim_model, dim_model)


class RelPosPatch1dMultiHeadAttention(RelPos1dMultiHeadAttention):
    """nnet/attentions.py:325-382: zero-pad to a multiple of P, avg-pool, attend over patches, nearest up-sample, slice."""

    def __init__(self, dim_model, num_heads, patch_size, num_pos_embeddings, attn_drop_rate, weight_init="scaled_uniform", bias_init="zeros", output_proj=True):
        super().__init__(dim_model, num_heads, num_pos_embeddings, attn_drop_rate, weight_init=weight_init, bias_init=bias_init, output_proj=output_proj)
        self.patch_size = patch_size


class Mask(nn.Module):
    """Binary mask, 1 = keep (nnet/attentions.py:656-733).  Without context limits it only encodes key padding and the conformer stack hands the lengths straight
    to the attention kernels (no (B,1,T,T) tensor, no per-sample host loop).  With left_context / right_context (streaming) the band
    j - i <= right_context, i - j <= left_context -- opened again on the first mask_start x mask_start block -- is materialised on the device, intersected with
    the key padding, and consumed by the dense-mask attention path; the stack strides it with the blocks exactly as the reference does."""

    def __init__(self, left_context=None, right_context=None, seq_len_axis=1, mask_start=0, unsqueeze_head=True):
        super().__init__()
        self.left_context, self.right_context, self.mask_start = left_context, right_context, mask_start
        self.seq_len_axis = [seq_len_axis] if isinstance(seq_len_axis, int) else seq_len_axis
        self.unsqueeze_head = unsqueeze_head

    @property
    def has_context(self):
        return self.left_context is not None or self.right_context is not None

    def forward(self, x, x_len=None):
        T = 1
        for ax in self.seq_len_axis:
            T *= x.size(ax)
        i = torch.arange(T, device=x.device)[:, None]
        j = torch.arange(T, device=x.device)[None, :]
        keep = torch.ones(T, T, dtype=torch.bool, device=x.device)
        if self.right_context is not None:
            keep &= (j - i) <= self.right_context
        if self.left_context is not None:
            keep &= (i - j) <= self.left_context
        if self.mask_start:
            keep |= (i < self.mask_start) & (j < self.mask_start)
        if x_len is None:
            m = keep[None]
        else:
            m = keep[None] & (j[None] < x_len.to(x.device)[:, None, None])
        m = m.to(x.dtype)
        return m[:, None] if self.unsqueeze_head else m


att_dict = {"RelPos1dMultiHeadAttention": RelPos1dMultiHeadAttention, "RelPosPatch1dMultiHeadAttention": RelPosPatch1dMultiHeadAttention}
