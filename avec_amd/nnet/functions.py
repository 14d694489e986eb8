"""Standalone forwards of leaf layers (registry contract) expressed with the same HIP launchers the fused composites use."""
import torch

from .. import ops
from .. import runtime as rt
from ..lib import ACT_NONE, ROWS_CONV_FWD, lib


class Conv2dFn(torch.autograd.Function):
    """layers.Conv2d.forward on a logical NCHW tensor: implicit-GEMM on its channels-last image; returns logical NCHW (channels-last strides)."""

    @staticmethod
    def forward(ctx, x, conv):
        rt.require_gpu(x)
        N, Cin, H, W = x.shape
        xa = x.permute(0, 2, 3, 1).to(rt.act_dtype()).contiguous()
        stride = conv.stride[0]
        Cout, KH, KW = conv.weight.shape[0], conv.weight.shape[2], conv.weight.shape[3]
        pad = (KH - 1) // 2 if conv.padding_type == "same" else 0
        OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        M = N * OH * OW
        sh = rt.shadow(conv.weight)
        y = ops.empty((M, Cout), rt.act_dtype(), xa)
        rows = ops.rows_conv(H, W, Cin, KH, KW, stride, pad, OH, OW)
        ops.gemm_nt(xa, sh.fwd, y, M, Cout, KH * KW * Cin, rows=rows, mode=ROWS_CONV_FWD, bias=conv.bias)
        ctx.saved = (xa, conv, rows, N, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, x.dtype)
        return y.view(N, OH, OW, Cout).permute(0, 3, 1, 2).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        from ..lib import ROWS_CONV_BWD
        xa, conv, rows, N, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, xdt = ctx.saved
        M = N * OH * OW
        dya = dy.permute(0, 2, 3, 1).to(rt.act_dtype()).contiguous().view(M, Cout)
        if conv.bias is not None:
            ops.colsum(dya, Cout, ops.grad_of(conv.bias), M, Cout)
        sh = rt.shadow(conv.weight)
        ops.gemm_tn(dya, xa, ops.grad_of(conv.weight), M, Cout, KH * KW * Cin, q_rows=rows, q_mode=ROWS_CONV_FWD)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.empty((N * H * W, Cin), rt.act_dtype(), dya)
            ops.gemm_nt(dya, sh.bwd, dx, N * H * W, Cin, KH * KW * Cout, rows=ops.rows_conv(H, W, Cout, KH, KW, stride, pad, OH, OW), mode=ROWS_CONV_BWD)
            dx = dx.view(N, H, W, Cin).permute(0, 3, 1, 2).to(xdt)
        return dx, None


def conv2d_module_forward(conv, x):
    if conv.channels_last:
        x = x.permute(0, 3, 1, 2)
    if not hasattr(conv.weight, "_avec_shadow") or conv.mask is not None:
        raise RuntimeError("standalone Conv2d with in_channels=%d is not part of the HIP hot path (the Cin=1 audio stem runs fused in "
                           "AudioEfficientConformerEncoder)" % conv.in_channels)
    y = Conv2dFn.apply(x, conv)
    return y.permute(0, 2, 3, 1) if conv.channels_last else y


class BatchNormFn(torch.autograd.Function):
    """BatchNorm{1,2,3}d.forward on a logical (N,C,...) tensor via its channels-last [M][C] image."""

    @staticmethod
    def forward(ctx, x, bn, training):
        rt.require_gpu(x)
        nd = x.dim()
        perm = (0,) + tuple(range(2, nd)) + (1,)
        xa = x.permute(perm).to(rt.act_dtype()).contiguous()
        C = xa.shape[-1]
        M = xa.numel() // C
        st = ops.BNState(C, xa)
        if training:
            lib.bn_stats(rt.dt(), xa.data_ptr(), st.stats.data_ptr(), M, C, rt.stream())
        cp = ops.bn_finalize(bn, st, M, training)
        out = torch.empty_like(xa)
        lib.bn_apply_fwd(rt.dt(), xa.data_ptr(), st.ss.data_ptr(), None, ACT_NONE, out.data_ptr(), M, C, rt.stream())
        ctx.saved = (xa, st, cp, bn, training, M, C, perm, x.dtype)
        inv = [0] * nd
        for i, p in enumerate(perm):
            inv[p] = i
        ctx.inv = inv
        return out.permute(inv).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xa, st, cp, bn, training, M, C, perm, xdt = ctx.saved
        dya = dy.permute(perm).to(rt.act_dtype()).contiguous()
        if training:
            dx, _ = ops.bn_backward(bn, st, cp, M, dya, xa, None, ACT_NONE, M)
        else:
            dx = ops._bn_eval_backward(bn, st, dya, xa, ACT_NONE, M)
        return dx.view(xa.shape).permute(ctx.inv).to(xdt), None, None


def batchnorm_module_forward(bn, x):
    training = bn.training and not bn.frozen
    if bn.channels_last:
        x = x.movedim(-1, 1)
    y = BatchNormFn.apply(x, bn, training)
    return y.movedim(1, -1) if bn.channels_last else y
