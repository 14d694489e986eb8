"""Parameter-holding layers with the reference's constructor signatures and state_dict keys (nnet/layers.py).

Storage is chosen for the MI355X kernels while the logical (PyTorch) shapes are unchanged:
  Conv2d  weight (Cout,Cin,KH,KW) lives channels-last  = physical [Cout][KH][KW][Cin]  (implicit-GEMM K order)
  Conv1d  depthwise weight (C,1,K) lives tap-major      = physical [K][C]
The GEMM-shaped weights register compute-dtype "shadows" (runtime.register_weight) in the two layouts the
NT MFMA kernel consumes (forward and backward-data)."""
import torch
import torch.nn as nn

from .. import ops
from .. import runtime as rt
from .initializations import apply_init


class Linear(nn.Linear):
    """nnet/layers.py:29-76"""

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None, weight_init="default", bias_init="default"):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        apply_init(self.weight, weight_init)
        apply_init(self.bias, bias_init)
        rt.register_weight(self.weight, out_features, 1, in_features)

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


def _same_pad(k):
    return ((k - 1) // 2, k // 2)


class Conv1d(nn.Conv1d):
    """nnet/layers.py:82-198.  On the hot path: pointwise (k=1, optionally strided) = GEMM; depthwise (groups=C) is executed
    fused with GLU by ConvolutionModule."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=True, padding_mode="zeros",
                 device=None, dtype=None, padding="same", channels_last=False, weight_init="default", bias_init="default", mask=None):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=0 if isinstance(padding, str) else padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode, device=device, dtype=dtype)
        self.padding_type = padding if isinstance(padding, str) else "valid"
        assert self.padding_type in ("valid", "same", "causal")
        self.channels_last = channels_last
        apply_init(self.weight, weight_init)
        apply_init(self.bias, bias_init)
        self.register_buffer("mask", mask)
        self.depthwise = groups == in_channels == out_channels and groups > 1
        if self.depthwise:
            w = self.weight.data          # (C,1,K) -> physical [K][C]
            self.weight.data = w.permute(2, 1, 0).contiguous().permute(2, 1, 0)
        else:
            assert groups == 1, "grouped (non-depthwise) Conv1d is not on the hot path"
            if self.kernel_size[0] == 1:
                rt.register_weight(self.weight, out_channels, 1, in_channels)

    def forward(self, x):
        if self.depthwise and self.mask is None:            # on the hot path it runs fused inside ConvolutionModule; alone: the same kernels with the GLU gate pinned to 1
            if self.dilation[0] != 1 or self.padding_type not in ("same", "causal"):
                raise RuntimeError("depthwise Conv1d on the HIP path: dilation 1 and padding 'same' or 'causal' (got dilation %d, padding %r); the fused GLU + depthwise kernels "
                                   "take neither a dilation nor an integer padding" % (self.dilation[0], self.padding_type))
            from .functions import DepthwiseConv1dFn
            if not self.channels_last:
                x = x.transpose(1, 2)
            y = DepthwiseConv1dFn.apply(x, self.weight, self)
            return y if self.channels_last else y.transpose(1, 2)
        if self.kernel_size[0] != 1 or self.mask is not None:
            raise RuntimeError("Conv1d(k=%d, groups=%d) on the HIP path: pointwise (k = 1) or depthwise (groups == channels)" % (self.kernel_size[0], self.groups))
        if not self.channels_last:
            x = x.transpose(1, 2)
        if self.stride[0] > 1:
            x = x[:, ::self.stride[0]]
        y = ops.linear(x, self.weight, self.bias)
        return y if self.channels_last else y.transpose(1, 2)


class Conv2d(nn.Conv2d):
    """nnet/layers.py:200-324 ("same" = explicit zero pad ((k-1)//2, k//2), folded into the implicit-GEMM loader)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=True, padding_mode="zeros",
                 device=None, dtype=None, padding="same", channels_last=False, weight_init="default", bias_init="default", mask=None):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=0 if isinstance(padding, str) else padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode, device=device, dtype=dtype)
        assert groups == 1 and self.dilation == (1, 1)
        self.padding_type = padding if isinstance(padding, str) else "valid"
        self.channels_last = channels_last
        apply_init(self.weight, weight_init)
        apply_init(self.bias, bias_init)
        self.register_buffer("mask", mask)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        if in_channels % 8 == 0:
            rt.register_weight(self.weight, out_channels, self.kernel_size[0] * self.kernel_size[1], in_channels)

    def forward(self, x):
        from .functions import conv2d_module_forward
        return conv2d_module_forward(self, x)


class Conv3d(nn.Conv3d):
    """nnet/layers.py:326-503.  Only the Cin=1 (5,7,7)/(1,2,2) stem shape is on the hot path (executed by the visual front-end)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=True, padding_mode="zeros",
                 device=None, dtype=None, padding="same", channels_last=False, weight_init="default", bias_init="default", mask=None):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=0 if isinstance(padding, str) else padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode, device=device, dtype=dtype)
        self.padding_type = padding if isinstance(padding, str) else "valid"
        self.channels_last = channels_last
        apply_init(self.weight, weight_init)
        apply_init(self.bias, bias_init)
        self.register_buffer("mask", mask)
        if in_channels == 1:
            k = self.kernel_size
            kk = k[0] * k[1] * k[2]
            rt.register_weight(self.weight, out_channels, 1, kk, need_bwd=False, c_pad=(kk + 7) // 8 * 8)

    def forward(self, x):
        from .functions import conv3d_module_forward
        return conv3d_module_forward(self, x)


class MaxPool3d(nn.MaxPool3d):
    """nnet/layers.py:839-915 (kept for the module tree; the stem max-pool is fused with BatchNorm3d+ReLU)."""

    def __init__(self, kernel_size, stride=None, dilation=1, return_indices=False, ceil_mode=False, padding="same", channels_last=False):
        super().__init__(kernel_size=kernel_size, stride=stride, padding=0, dilation=dilation, return_indices=return_indices, ceil_mode=ceil_mode)
        self.padding_type = padding
        self.channels_last = channels_last

    def forward(self, x):
        from .functions import maxpool3d_module_forward
        return maxpool3d_module_forward(self, x)


class AvgPool1d(nn.AvgPool1d):
    def __init__(self, kernel_size, stride=None, padding=0, ceil_mode=False, count_include_pad=True, channels_last=False):
        super().__init__(kernel_size=kernel_size, stride=stride, padding=padding, ceil_mode=ceil_mode, count_include_pad=count_include_pad)
        self.channels_last = channels_last

    def forward(self, x):
        from .functions import avgpool1d_module_forward
        return avgpool1d_module_forward(self, x)


class Upsample(nn.Upsample):
    def __init__(self, size=None, scale_factor=None, mode="nearest", align_corners=None, recompute_scale_factor=None, channels_last=False):
        super().__init__(size=size, scale_factor=scale_factor, mode=mode, align_corners=align_corners, recompute_scale_factor=recompute_scale_factor)
        self.channels_last = channels_last

    def forward(self, x):
        from .functions import upsample_module_forward
        return upsample_module_forward(self, x)


class Dropout(nn.Dropout):
    def __init__(self, p=0.5, inplace=False):
        super().__init__(p=p, inplace=inplace)
        self.rng_stream = rt.new_stream_id()

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        return ops.DropoutFn.apply(x, self.p, self.rng_stream)


# ---- pure view helpers (no kernels) ----------------------------------------------------------
class PermuteChannels(nn.Module):
    def __init__(self, to_last=True, num_dims=None, make_contiguous=False):
        super().__init__()
        self.to_last, self.num_dims, self.make_contiguous = to_last, num_dims, make_contiguous

    def forward(self, x):
        n = x.dim() - 2 if self.num_dims is None else self.num_dims
        dims = (0,) + tuple(range(2, n + 2)) + (1,) if self.to_last else (0, n + 1) + tuple(range(1, n + 1))
        x = x.permute(dims)
        return x.contiguous() if self.make_contiguous else x


class Flatten(nn.Flatten):
    pass


class Transpose(nn.Module):
    def __init__(self, dim0, dim1):
        super().__init__()
        self.dim0, self.dim1 = dim0, dim1

    def forward(self, x):
        return x.transpose(self.dim0, self.dim1)


class Permute(nn.Module):
    def __init__(self, dims, make_contiguous=False):
        super().__init__()
        self.dims, self.make_contiguous = dims, make_contiguous

    def forward(self, x):
        x = x.permute(self.dims)
        return x.contiguous() if self.make_contiguous else x


class Reshape(nn.Module):
    def __init__(self, shape, include_batch=True):
        super().__init__()
        self.shape, self.include_batch = tuple(shape), include_batch

    def forward(self, x):
        return x.reshape(self.shape) if self.include_batch else x.reshape(x.size()[0:1] + self.shape)


class Unsqueeze(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return x.unsqueeze(dim=self.dim)


class GlobalAvgPool2d(nn.Module):
    """nnet/layers.py:1328-1342; executed by the ResNet head on channels-last activations."""

    def __init__(self, dim=(2, 3), keepdim=False):
        super().__init__()
        self.dim, self.keepdim = dim, keepdim

    def forward(self, x, mask=None):
        from .functions import global_avgpool2d_module_forward
        return global_avgpool2d_module_forward(self, x, mask)


layer_dict = {
    "Linear": Linear, "Conv1d": Conv1d, "Conv2d": Conv2d, "Conv3d": Conv3d, "MaxPool3d": MaxPool3d, "Dropout": Dropout,
    "Flatten": Flatten, "Transpose": Transpose, "Permute": Permute, "Reshape": Reshape, "Unsqueeze": Unsqueeze,
    "GlobalAvgPool2d": GlobalAvgPool2d,
}
