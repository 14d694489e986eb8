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


class RelPosMultiHeadSelfAttention(MultiHeadAttention):
    """nnet/attentions.py:384-554 (Transformer-XL form): scores = ((Q + u) K^T + rel_to_abs((Q + v) E^T)) / sqrt(d), optional key/value cache (`hidden`).
    Here = the grouped class below with group_size 1."""

    def __init__(self, dim_model, num_heads, attn_drop_rate, max_pos_encoding, weight_init="scaled_uniform", bias_init="zeros", output_proj=True, causal=False,
                 group_size=1):
        super().__init__(dim_model, num_heads, attn_drop_rate, weight_init=weight_init, bias_init=bias_init, output_proj=output_proj)
        assert not causal, "causal Transformer-XL attention is not on this path (the encoders pass causal=False, nnet/networks.py:330)"
        self.pos_layer = layers.Linear(dim_model, dim_model)
        self.causal = causal
        self.u = nn.Parameter(torch.zeros(dim_model))          # content bias
        self.v = nn.Parameter(torch.zeros(dim_model))          # position bias
        self.group_size = group_size
        self.dim_head = (group_size * dim_model) // num_heads
        self.rel_pos_enc = embeddings.GroupedRelativeSinusoidalPositionalEncoding(max_pos_encoding, dim_model, group_size, causal)

    # The u / v biases and the frame grouping are expressed on the operands of the shared attention core (ops.RelPosCoreFn):
    #   (Q + u) K^T + (Q + v) E^T = [Q, 1] ([K, u.K] + [E, v.E])^T      -> one extra channel per head carries u.K_j resp. v.E_r;
    #   a grouped token = G consecutive frames, its heads = slices of the G*D vector (a pure reshape, nnet/attentions.py:613-619).
    def _operands(self, Q, K, V, T, Th):
        """Q (B,T,D), K / V (B,Th+T,D) fp32, already zero-padded to multiples of G -> augmented per-head operands and the position operand"""
        B, G, H, dh = Q.shape[0], self.group_size, self.num_heads, self.dim_head
        Tg, Tkg = T // G, (Th + T) // G
        d1 = dh + 1 + ((dh + 1) % 2)                          # + the bias channel, padded to an even width
        u, v = self.u.repeat(G).view(1, 1, H, dh), self.v.repeat(G).view(1, H, dh)
        Qg, Kg, Vg = Q.reshape(B, Tg, H, dh), K.reshape(B, Tkg, H, dh), V.reshape(B, Tkg, H, dh)
        zq, zk = Qg.new_zeros(B, Tg, H, d1 - dh - 1), Kg.new_zeros(B, Tkg, H, d1 - dh - 1)
        qa = torch.cat([Qg, torch.ones_like(Qg[..., :1]), zq], -1)
        ka = torch.cat([Kg, (Kg * u).sum(-1, keepdim=True), zk], -1)
        va = torch.cat([Vg, Vg.new_zeros(B, Tkg, H, d1 - dh)], -1)
        pe = self.rel_pos_enc(1, T, Th, device=Q.device)[0]                                   # (Th + 2T - G, D) fp32
        E = ops.linear(pe, self.pos_layer.weight, self.pos_layer.bias)                        # fp32
        Eg = E.reshape(-1, H, dh)                                                             # (Tkg + Tg - 1, H, dh)
        ea = torch.cat([Eg, (Eg * v).sum(-1, keepdim=True), Eg.new_zeros(Eg.shape[0], H, d1 - dh - 1)], -1)
        return qa, ka, va, ea, d1

    def _attend(self, h, mask, lengths, hidden=None, want_w=False):
        """h: (B,T,D) fp32 (already normalised) -> (attention output after the output Linear (B,T,D) fp32, probabilities or None, updated cache)"""
        B, T, D = h.shape
        G, H, dh = self.group_size, self.num_heads, self.dim_head
        adt = rt.act_dtype()
        Q = ops.linear(h, self.query_layer.weight, self.query_layer.bias)
        K = ops.linear(h, self.key_layer.weight, self.key_layer.bias)
        V = ops.linear(h, self.value_layer.weight, self.value_layer.bias)
        new_hidden = None
        if hidden:                                             # nnet/attentions.py:588-600: the cache keeps every frame; the attention drops the first Th % G of them
            Kh, Vh = torch.cat([hidden["K"].to(K), K], 1), torch.cat([hidden["V"].to(V), V], 1)
            cut = hidden["K"].shape[1] % G
            K, V = torch.cat([hidden["K"].to(K)[:, cut:], K], 1), torch.cat([hidden["V"].to(V)[:, cut:], V], 1)
            new_hidden = {"K": Kh.detach(), "V": Vh.detach()}
        elif want_w:
            new_hidden = {"K": K.detach(), "V": V.detach()}
        pq, pk = (-T) % G, (-K.shape[1]) % G                   # MultiHeadAttention.pad (nnet/attentions.py:140-171): zeros AFTER the projections
        F_ = torch.nn.functional
        Q, K, V = F_.pad(Q, (0, 0, 0, pq)), F_.pad(K, (0, 0, 0, pk)), F_.pad(V, (0, 0, 0, pk))
        Tp, Tkp = T + pq, K.shape[1]
        Tg, Tkg = Tp // G, Tkp // G
        if lengths is not None:
            lens_g, mask_g = torch.div(lengths.to(device=h.device, dtype=torch.int64) + (G - 1), G, rounding_mode="floor").contiguous(), None    # key group j' is visible iff frame G*j' is
        elif mask is not None:
            m = mask.reshape(mask.shape[0], 1, mask.shape[-2], mask.shape[-1]).float()
            m = F_.pad(m, (0, Tkp - m.shape[-1], 0, Tp - m.shape[-2]) if m.shape[-2] > 1 else (0, Tkp - m.shape[-1]), value=0.0)
            if m.shape[-2] == 1:
                m = m.expand(-1, -1, Tp, -1)
            lens_g, mask_g = None, m[:, 0, ::G, ::G].contiguous()
        elif pk:
            lens_g, mask_g = None, h.new_zeros(1, Tg, Tkg)     # the reference builds an all-zero mask here (every key masked: uniform attention) -- reproduced
        else:
            lens_g = mask_g = None
        qa, ka, va, ea, d1 = self._operands(Q, K, V, Tp, Tkp - Tp)
        scale = 1.0 / dh ** 0.5
        att_w = None
        if Tkg == Tg and not want_w:
            qkv = torch.cat([qa.reshape(B * Tg, H * d1), ka.reshape(B * Tg, H * d1), va.reshape(B * Tg, H * d1)], 1).to(adt).contiguous()
            o = ops.RelPosCoreFn.apply(qkv, ea.reshape(-1, H * d1).to(adt).contiguous(), lens_g, 1, mask_g, B, H, Tg, d1, scale)
        else:
            assert not torch.is_grad_enabled() or not h.requires_grad, "a key/value cache (hidden) / returned attention weights are inference features"
            o, att_w = ops.relpos_core_infer(qa.reshape(B * Tg, H * d1).to(adt).contiguous(), ka.reshape(B * Tkg, H * d1).to(adt).contiguous(),
                                             va.reshape(B * Tkg, H * d1).to(adt).contiguous(), ea.reshape(-1, H * d1).to(adt).contiguous(),
                                             lens_g, mask_g, B, H, Tg, Tkg, d1, scale, want_probs=want_w)
        o = o.float().view(B, Tg, H, d1)[..., :dh].reshape(B, Tp, D)[:, :T]
        return ops.linear(o, self.output_layer.weight, self.output_layer.bias), att_w, new_hidden

    def fused(self, x, ln, mask, lengths, drop_p, sid, residual):
        """y = [x +] Drop(attention(LN(x))) -- the AttentionModule entry (nnet/modules.py:320-339)"""
        h = ops.LayerNormFn.apply(x.float(), ln.weight, ln.bias, ln.eps) if ln is not None else x.float()
        o, _, _ = self._attend(h, None if lengths is not None else mask, lengths)
        if drop_p > 0:
            o = ops.DropoutFn.apply(o, drop_p, sid)
        return x.float() + o if residual else o

    def forwardQKV(self, Q, K, V, mask=None, return_att_w=False, hidden=None):
        assert Q is K and K is V, "only self-attention is on the hot path"
        o, att_w, new_hidden = self._attend(Q.float(), mask, None, hidden=hidden, want_w=return_att_w)
        return (o, att_w, new_hidden) if return_att_w else o

    def forward(self, x, mask=None, return_att_w=False, hidden=None):
        return self.forwardQKV(x, x, x, mask, return_att_w, hidden)


class GroupedRelPosMultiHeadSelfAttention(RelPosMultiHeadSelfAttention):
    """nnet/attentions.py:556-650: G consecutive frames form one attention token of head width G*D/H (G = 3, D = 180, H = 4: 135); the mask is sub-sampled
    [::G, ::G]; zero padding to a multiple of G after the projections."""

    def __init__(self, dim_model, num_heads, attn_drop_rate, max_pos_encoding, group_size, causal, weight_init="scaled_uniform", bias_init="zeros", output_proj=True):
        super().__init__(dim_model, num_heads, attn_drop_rate, max_pos_encoding, weight_init=weight_init, bias_init=bias_init, output_proj=output_proj, causal=causal,
                         group_size=group_size)


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


att_dict = {"RelPos1dMultiHeadAttention": RelPos1dMultiHeadAttention, "RelPosPatch1dMultiHeadAttention": RelPosPatch1dMultiHeadAttention,
            "RelPosMultiHeadSelfAttention": RelPosMultiHeadSelfAttention, "GroupedRelPosMultiHeadSelfAttention": GroupedRelPosMultiHeadSelfAttention}
