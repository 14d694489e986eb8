"""Composite modules (nnet/modules.py).  The sub-module tree (hence state_dict keys) is the reference's; `forward` is one fused HIP
sequence per module instead of one ATen op per child."""
import torch
import torch.nn as nn

from .. import ops
from .. import runtime as rt
from . import activations, attentions, layers, normalizations


class LengthMask:
    """Key-padding mask carried as per-utterance lengths (what Mask() encodes, nnet/attentions.py:682-733)."""

    def __init__(self, lengths):
        self.lengths = lengths

    def strided(self, s):
        return LengthMask(ops.len_affine(self.lengths, 1, s, 1))


def _act_params(act_fun):
    if isinstance(act_fun, dict):
        return activations.act_dict[act_fun["class"]], act_fun["params"]
    return activations.act_dict[act_fun], {}


def _norm_params(norm):
    if isinstance(norm, dict):
        return normalizations.norm_dict[norm["class"]], norm["params"]
    return normalizations.norm_dict[norm], {}


class ConvNeuralNetwork(nn.Module):
    """nnet/modules.py:70-130.  Used for the two stems; their forward is executed by the encoders' fused front-ends."""

    def __init__(self, dim_input, dim_layers, kernel_size, strides=1, norm=None, act_fun="ReLU", drop_rate=0.0, padding="same", dim=2,
                 channels_last=False, residual=False, weight_init="default", bias_init="default", bias=True):
        super().__init__()
        conv = {1: layers.Conv1d, 2: layers.Conv2d, 3: layers.Conv3d}[dim]
        act, act_kw = _act_params(act_fun)
        nrm, nrm_kw = _norm_params(norm)
        self.strides, self.residual = strides, residual
        dims = [dim_layers] if isinstance(dim_layers, int) else dim_layers
        pick = lambda v, i: v[i] if isinstance(v, list) else v
        self.layers = nn.ModuleList([nn.Sequential(
            conv(dim_input if i == 0 else dims[i - 1], dims[i], pick(kernel_size, i), stride=pick(strides, i), padding=pick(padding, i),
                 channels_last=channels_last, weight_init=weight_init, bias_init=bias_init, bias=bias),
            nrm(dims[i], **nrm_kw, channels_last=channels_last) if nrm is not nn.Identity else nn.Identity(),
            act(**act_kw),
            nn.Dropout(drop_rate) if drop_rate > 0 else nn.Identity()) for i in range(len(dims))])

    def forward(self, x, x_len=None):
        """nnet/modules.py:115-130: conv -> norm -> activation -> dropout per layer (+ residual); every layer halves the lengths (the reference hard-codes 2).
        On the hot path the two stems run as fused sequences (ops.AudioStemFn / ops.VideoStemFn, called by the encoders); called on its own the stack runs layer by
        layer through the stand-alone layer forwards (same kernels, more launches)."""
        for layer in self.layers:
            x = x + layer(x) if self.residual else layer(x)
            if x_len is not None:
                x_len = torch.div(x_len - 1, 2, rounding_mode="floor") + 1
        return x if x_len is None else (x, x_len)


class FeedForwardModule(nn.Module):
    """nnet/modules.py:257-289: LN -> Linear(D,4D) -> Swish -> Dropout -> Linear(4D,D) -> Dropout."""

    def __init__(self, dim_model, dim_ffn, drop_rate, act_fun, inner_dropout, prenorm=True, weight_init="default", bias_init="default"):
        super().__init__()
        assert prenorm and act_fun == "Swish", "hot-path FFN is pre-norm + Swish (nnet/blocks.py:229-236)"
        self.layers = nn.Sequential(
            nn.LayerNorm(dim_model, eps=1e-6),
            layers.Linear(dim_model, dim_ffn, weight_init=weight_init, bias_init=bias_init),
            activations.act_dict[act_fun](),
            nn.Dropout(p=drop_rate) if inner_dropout else nn.Identity(),
            layers.Linear(dim_ffn, dim_model, weight_init=weight_init, bias_init=bias_init),
            nn.Dropout(p=drop_rate))
        self.drop_rate, self.inner_dropout = drop_rate, inner_dropout
        self.sid1, self.sid2 = rt.new_stream_id(), rt.new_stream_id()

    def residual_forward(self, x, alpha):
        """x + alpha * FFN(x) in one fused sequence (the macaron half-step of nnet/blocks.py:292,301)."""
        ln, l1, l2 = self.layers[0], self.layers[1], self.layers[4]
        p = self.drop_rate if self.training else 0.0
        assert self.inner_dropout or p == 0.0
        return ops.FeedForwardFn.apply(x, ln.weight, ln.bias, l1.weight, l1.bias, l2.weight, l2.bias, ln.eps, alpha, p, self.sid1, self.sid2)

    def forward(self, x):
        return self.residual_forward(x, 1.0) - x


class AttentionModule(nn.Module):
    """nnet/modules.py:291-339: LN -> attention.forwardQKV -> Dropout (+ residual)."""

    def __init__(self, dim_model, att_params, drop_rate, norm={"class": "LayerNorm", "params": {"eps": 1e-6}}, residual=True, channels_last=True):
        super().__init__()
        nrm, kw = _norm_params(norm)
        self.norm = nrm(dim_model, **kw, channels_last=channels_last)
        self.attention = attentions.att_dict[att_params["class"]](dim_model=dim_model, **att_params["params"])
        self.dropout = nn.Dropout(drop_rate)
        self.residual = residual
        self.sid = rt.new_stream_id()

    def forward(self, x, x_cross=None, mask=None, add_residual=None):
        assert x_cross is None, "cross-attention is not on the hot path"
        lengths = None
        if isinstance(mask, LengthMask):
            lengths, mask = mask.lengths, None
        p = self.dropout.p if self.training else 0.0
        res = self.residual if add_residual is None else add_residual
        return self.attention.fused(x, self.norm, mask, lengths, p, self.sid, res)


class ConvolutionModule(nn.Module):
    """nnet/modules.py:341-385: LN -> pointwise(D->2D') -> GLU -> depthwise(k, stride) -> BatchNorm -> Swish -> pointwise -> Dropout."""

    def __init__(self, dim_model, dim_expand, drop_rate, stride, act_fun="Swish", conv_params={"class": "Conv2d", "params": {"padding": "same", "kernel_size": 3}},
                 channels_last=False, batch_norm=True):
        super().__init__()
        assert conv_params["class"] == "Conv1d" and channels_last and batch_norm and act_fun == "Swish", "hot-path conv module: channels-last Conv1d + BatchNorm1d + Swish"
        self.layers = nn.Sequential(
            normalizations.LayerNorm(dim_model, channels_last=True, eps=1e-6),
            layers.Conv1d(dim_model, 2 * dim_expand, kernel_size=1, channels_last=True),
            activations.GLU(dim=-1),
            layers.Conv1d(dim_expand, dim_expand, stride=stride, groups=dim_expand, channels_last=True, **conv_params["params"]),
            normalizations.BatchNorm1d(dim_expand, channels_last=True),
            activations.act_dict[act_fun](),
            layers.Conv1d(dim_expand, dim_expand, kernel_size=1, channels_last=True),
            nn.Dropout(p=drop_rate))
        self.sid = rt.new_stream_id()

    def residual_forward(self, x, res_conv):
        """R(x) + ConvModule(x) with R = identity or the strided k=1 conv of the block (nnet/blocks.py:273-277,298)."""
        p = self.layers[7].p if self.training else 0.0
        return ops.ConvModuleFn.apply(x, self.layers[1].weight, self, res_conv, p, self.sid, self.training)

    def forward(self, x):
        assert self.layers[3].stride[0] == 1 and self.layers[1].in_channels == self.layers[6].out_channels
        return self.residual_forward(x, None) - x


class InterCTCResModule(nn.Module):
    """nnet/modules.py:387-400"""

    def __init__(self, dim_model, vocab_size):
        super().__init__()
        self.proj_1 = layers.Linear(dim_model, vocab_size)
        self.proj_2 = layers.Linear(vocab_size, dim_model)

    def forward(self, x):
        return ops.InterCTCFn.apply(x, self.proj_1.weight, self.proj_1.bias, self.proj_2.weight, self.proj_2.bias)


class FusionModule(nn.Module):
    """nnet/modules.py:402-426"""

    def __init__(self, a_dim_model=360, v_dim_model=360, f_dim_model=360, ff_ratio=4):
        super().__init__()
        self.layers = nn.Sequential(
            layers.Linear(a_dim_model + v_dim_model, ff_ratio * f_dim_model),
            activations.Swish(),
            layers.Linear(ff_ratio * f_dim_model, f_dim_model))

    def forward(self, audio, video):
        l1, l2 = self.layers[0], self.layers[2]
        return ops.FusionFn.apply(audio, video, l1.weight, l1.bias, l2.weight, l2.bias)
