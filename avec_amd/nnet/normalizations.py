"""Normalisation layers (nnet/normalizations.py): same constructors / state_dict keys; forward = HIP kernels."""
import torch
import torch.nn as nn

from .. import ops
from .initializations import apply_init


class LayerNorm(nn.LayerNorm):
    """nnet/normalizations.py:27-40"""

    def __init__(self, normalized_shape, eps=1e-05, elementwise_affine=True, device=None, dtype=None, channels_last=True):
        super().__init__(normalized_shape, eps=eps, elementwise_affine=elementwise_affine, device=device, dtype=dtype)
        self.channels_last = channels_last

    def forward(self, x):
        if not self.channels_last:
            x = x.transpose(1, -1)
        y = ops.LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return y if self.channels_last else y.transpose(1, -1)


def torch_layer_norm_forward(ln, x, nxt=None):
    """nn.LayerNorm instances created by the composites (FeedForwardModule / ConformerBlock) use the same kernel.
    nxt: the LayerNorm that reads the result next (the following block's first pre-norm): its output comes out of the same launch (ops.LN_PAIR)."""
    return ops.LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, None if nxt is None else (nxt.weight, nxt.bias, nxt.eps))


class _BatchNormMixin:
    def _setup(self, channels_last, weight_init, bias_init, frozen):
        self.frozen = frozen
        self.channels_last = channels_last
        if self.affine:
            apply_init(self.weight, weight_init)
            apply_init(self.bias, bias_init)

    def forward(self, x):
        from .functions import batchnorm_module_forward
        return batchnorm_module_forward(self, x)


class BatchNorm1d(_BatchNormMixin, nn.BatchNorm1d):
    def __init__(self, num_features, eps=1e-05, momentum=0.1, affine=True, track_running_stats=True, device=None, dtype=None,
                 channels_last=False, weight_init="default", bias_init="default", frozen=False):
        nn.BatchNorm1d.__init__(self, num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats, device=device, dtype=dtype)
        self._setup(channels_last, weight_init, bias_init, frozen)


class BatchNorm2d(_BatchNormMixin, nn.BatchNorm2d):
    def __init__(self, num_features, eps=1e-05, momentum=0.1, affine=True, track_running_stats=True, device=None, dtype=None,
                 channels_last=False, weight_init="default", bias_init="default", frozen=False):
        nn.BatchNorm2d.__init__(self, num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats, device=device, dtype=dtype)
        self._setup(channels_last, weight_init, bias_init, frozen)


class BatchNorm3d(_BatchNormMixin, nn.BatchNorm3d):
    def __init__(self, num_features, eps=1e-05, momentum=0.1, affine=True, track_running_stats=True, device=None, dtype=None,
                 channels_last=False, weight_init="default", bias_init="default", frozen=False):
        nn.BatchNorm3d.__init__(self, num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats, device=device, dtype=dtype)
        self._setup(channels_last, weight_init, bias_init, frozen)


class SyncBatchNorm:
    """The reference converts every BatchNorm to torch's SyncBatchNorm under DDP (nnet/normalizations.py:172-249, nnet/model.py:61).
    Here synchronisation is a property of the statistics exchange, not of the module class: `convert_sync_batchnorm` switches the
    engine to all-reduce the per-channel (sum, sumsq, count) / (sum dy, sum dy*xhat) vectors over RCCL and returns the module unchanged
    (state_dict keys identical to the reference's converted model)."""

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        from .. import runtime as rt
        rt.set_sync_batchnorm(True)
        return module


norm_dict = {None: nn.Identity, "LayerNorm": LayerNorm, "BatchNorm1d": BatchNorm1d, "BatchNorm2d": BatchNorm2d, "BatchNorm3d": BatchNorm3d}
