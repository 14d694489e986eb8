"""Activation registry (nnet/activations.py:71-82).  Inside the fused composites (FeedForwardModule, ConvolutionModule,
FusionModule, ResNetBlock, stems) the activation runs in the producing kernel's epilogue; these classes carry the name."""
import torch.nn as nn


class _FusedOnly(nn.Module):
    fused_act = 0

    def forward(self, x):
        raise RuntimeError("%s is executed inside the fused HIP composites (FeedForwardModule, ConvolutionModule, FusionModule, "
                           "ResNetBlock, stems); a standalone elementwise launch is not part of the MI355X hot path" % type(self).__name__)


class Identity(nn.Module):
    fused_act = 0

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return x


class ReLU(_FusedOnly):
    fused_act = 2

    def __init__(self, inplace=False):
        super().__init__()


class Swish(_FusedOnly):
    fused_act = 1


class GLU(_FusedOnly):
    def __init__(self, dim=-1):
        super().__init__()
        self.dim = dim


act_dict = {None: Identity, "Identity": Identity, "ReLU": ReLU, "Swish": Swish, "GLU": GLU}
