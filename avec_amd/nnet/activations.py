"""Activations (nnet/activations.py:39-82).  Inside the fused composites (FeedForwardModule, ConvolutionModule, FusionModule, ResNetBlock, stems) the
activation runs in the producing kernel's epilogue and these classes only carry its id; called directly they run the stand-alone HIP kernel."""
import torch.nn as nn


class _FusedOnly(nn.Module):
    """`fused_act` is what the composites pass to the producing kernel's epilogue; called on its own the module runs the stand-alone kernel (avec_act_f32)."""
    fused_act = 0
    act_id = 0

    def forward(self, x):
        from .. import ops
        return ops.ActivationFn.apply(x, self.act_id, getattr(self, "dim", -1))


class Identity(nn.Module):
    fused_act = 0

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return x


class ReLU(_FusedOnly):
    fused_act = 2
    act_id = 2

    def __init__(self, inplace=False):
        super().__init__()


class Swish(_FusedOnly):
    fused_act = 1
    act_id = 1


class GLU(_FusedOnly):
    act_id = 3

    def __init__(self, dim=-1):
        super().__init__()
        self.dim = dim


act_dict = {None: Identity, "Identity": Identity, "ReLU": ReLU, "Swish": Swish, "GLU": GLU}
