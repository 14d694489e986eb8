"""Conformer / ResNet blocks (nnet/blocks.py)."""
import torch
import torch.nn as nn

from .. import ops
from . import activations, layers, modules, normalizations


class ResNetBlock(nn.Module):
    """Basic residual block (nnet/blocks.py:29-91): conv3x3(s)-BN-ReLU-conv3x3-BN (+ 1x1(s)-BN shortcut) -> ReLU.
    forward takes/returns the channels-last image tensor (N,H,W,C) in the activation dtype."""

    def __init__(self, in_features, out_features, kernel_size, stride, norm="BatchNorm2d", act_fun="ReLU", dim=2, channels_last=False,
                 weight_init="he_normal", bias_init="zeros", bias=False, joined_post_act=False, padding="same"):
        super().__init__()
        assert dim == 2 and not bias and joined_post_act and act_fun == "ReLU" and norm == "BatchNorm2d" and padding == "same", \
            "hot-path ResNet block: bias-free 2d convs, BatchNorm2d, joined post-activation ReLU (nnet/networks.py:119-126)"
        act = activations.act_dict[act_fun]
        mk = lambda cin, k, s: layers.Conv2d(cin, out_features, kernel_size=k, stride=s, channels_last=channels_last, bias=False,
                                             weight_init=weight_init, bias_init=bias_init, **({"padding": padding} if k != 1 else {}))
        self.layers = nn.Sequential(mk(in_features, kernel_size, stride), normalizations.BatchNorm2d(out_features, channels_last=channels_last), act(),
                                    mk(out_features, kernel_size, 1), normalizations.BatchNorm2d(out_features, channels_last=channels_last), nn.Identity())
        self.joined_post_act = act()
        s = stride if isinstance(stride, int) else stride[0]
        if s > 1 or in_features != out_features:
            self.residual = nn.Sequential(mk(in_features, 1, stride), normalizations.BatchNorm2d(out_features, channels_last=channels_last))
        else:
            self.residual = nn.Identity()

    def forward_nhwc(self, x, chain=False):
        """chain: the output feeds ONLY the next ResNetBlock (lets that block's backward fold this block's BatchNorm-backward reduction into its epilogue)"""
        return ops.ResNetBlockFn.apply(x, self.layers[0].weight, self, self.training, chain)

    def forward(self, x):
        """logical NCHW in / out (reference interface)"""
        from .. import runtime as rt
        y = self.forward_nhwc(x.permute(0, 2, 3, 1).to(rt.act_dtype()).contiguous())
        return y.permute(0, 3, 1, 2)


class ConformerBlock(nn.Module):
    """nnet/blocks.py:208-306:  x += 1/2 FFN1;  x += MHSA;  x = conv_res(x) + Conv(x);  x += 1/2 FFN2;  x = LN(x)."""

    def __init__(self, dim_model, dim_expand, ff_ratio, att_params, drop_rate, conv_stride, conv_params, inner_dropout=True, act_fun="Swish",
                 batch_norm=True, block_norm=True):
        super().__init__()
        assert conv_params["class"] == "Conv1d", "hot-path conformer blocks use Conv1d convolution modules"
        self.ff_module1 = modules.FeedForwardModule(dim_model, dim_model * ff_ratio, drop_rate, act_fun, inner_dropout)
        self.self_att_module = modules.AttentionModule(dim_model, att_params, drop_rate, residual=False)
        self.conv_module = modules.ConvolutionModule(dim_model, dim_expand, drop_rate, conv_stride, act_fun=act_fun, conv_params=conv_params,
                                                     channels_last=True, batch_norm=batch_norm)
        self.ff_module2 = modules.FeedForwardModule(dim_expand, dim_expand * ff_ratio, drop_rate, act_fun, inner_dropout)
        self.norm = nn.LayerNorm(dim_expand, eps=1e-6) if block_norm else nn.Identity()
        if dim_model != dim_expand:
            self.conv_res = layers.Conv1d(dim_model, dim_expand, kernel_size=1, stride=conv_stride, channels_last=True)
        else:
            assert conv_stride == 1, "same-width strided blocks do not occur (and are dead code in the reference: no MaxPool1d in layer_dict)"
            self.conv_res = nn.Identity()
        self.stride = conv_stride

    def forward(self, x, mask=None):
        x = self.ff_module1.residual_forward(x, 0.5)
        x = self.self_att_module(x, mask=mask, add_residual=True)
        x = self.conv_module.residual_forward(x, None if isinstance(self.conv_res, nn.Identity) else self.conv_res)
        x = self.ff_module2.residual_forward(x, 0.5)
        if isinstance(self.norm, nn.LayerNorm):
            x = normalizations.torch_layer_norm_forward(self.norm, x, getattr(self, "_next_ln", None))
        return x


block_dict = {"ConformerBlock": ConformerBlock}
