"""Standalone forwards of leaf layers (registry contract) expressed with the same HIP launchers the fused composites use."""
import torch

from .. import ops
from .. import runtime as rt
from ..lib import ACT_NONE, ROWS_CONV_FWD, lib


class Conv2dFn(torch.autograd.Function):
    """layers.Conv2d.forward on a logical NCHW tensor: implicit-GEMM on its channels-last image; returns logical NCHW (channels-last strides)."""

    @staticmethod
    def forward(ctx, x, _anchor, conv):
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
        return dx, None, None


def conv2d_module_forward(conv, x):
    if conv.channels_last:
        x = x.permute(0, 3, 1, 2)
    if not hasattr(conv.weight, "_avec_shadow") or conv.mask is not None:
        raise RuntimeError("standalone Conv2d with in_channels=%d is not part of the HIP hot path (the Cin=1 audio stem runs fused in "
                           "AudioEfficientConformerEncoder)" % conv.in_channels)
    y = Conv2dFn.apply(x, conv.weight, conv)        # (the weight rides along so that the node exists when x carries no gradient)
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


# ---------------------------------------------------------------------------------------------------------------------------------------------------
# Stand-alone forwards of the layers the hot path only runs fused: MaxPool3d, AvgPool1d, Upsample, GlobalAvgPool2d, the depthwise
# Conv1d, the Conv3d stem shape.  Same kernels as the fused composites where they exist (patch pooling, global average, GLU + depthwise conv, stem im2col + GEMM),
# small dedicated kernels otherwise (csrc/standalone.hip).
# ---------------------------------------------------------------------------------------------------------------------------------------------------
class MaxPoolHWFn(torch.autograd.Function):
    """layers.MaxPool3d with a (1, KH, KW) window on a logical (N, C, T, H, W) tensor (nnet/layers.py:839-915: zero padding, then a valid max pool)"""

    @staticmethod
    def forward(ctx, x, k, s, pad0, pad1):
        rt.require_gpu(x)
        N, C, T, H, W = x.shape
        xa = x.permute(0, 2, 3, 4, 1).to(rt.act_dtype()).contiguous()                 # [N*T][H][W][C]
        OH, OW = (H + pad0[0] + pad1[0] - k[0]) // s[0] + 1, (W + pad0[1] + pad1[1] - k[1]) // s[1] + 1
        out = torch.empty((N, T, OH, OW, C), dtype=xa.dtype, device=x.device)
        idx = torch.empty((N, T, OH, OW, C), dtype=torch.uint8, device=x.device)
        lib.maxpool_hw_fwd(rt.dt(), xa.data_ptr(), out.data_ptr(), idx.data_ptr(), N * T, H, W, C, k[0], k[1], s[0], s[1], pad0[0], pad0[1], pad1[0], pad1[1], rt.stream())
        ctx.saved = (idx, (N, C, T, H, W), k, s, pad0, pad1, x.dtype)
        return out.permute(0, 4, 1, 2, 3).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        idx, (N, C, T, H, W), k, s, pad0, pad1, xdt = ctx.saved
        dya = dy.permute(0, 2, 3, 4, 1).to(rt.act_dtype()).contiguous()
        dx = torch.empty((N, T, H, W, C), dtype=dya.dtype, device=dy.device)
        lib.maxpool_hw_bwd(rt.dt(), dya.data_ptr(), idx.data_ptr(), dx.data_ptr(), N * T, H, W, C, k[0], k[1], s[0], s[1], pad0[0], pad0[1], pad1[0], pad1[1], rt.stream())
        return dx.permute(0, 4, 1, 2, 3).to(xdt), None, None, None, None


def maxpool3d_module_forward(mp, x):
    k = mp.kernel_size if isinstance(mp.kernel_size, tuple) else (mp.kernel_size,) * 3
    s = mp.stride if isinstance(mp.stride, tuple) else ((mp.stride,) * 3 if mp.stride is not None else k)
    if k[0] != 1 or s[0] != 1 or mp.dilation not in (1, (1, 1, 1)) or mp.return_indices or mp.ceil_mode or mp.padding_type == "causal":
        raise RuntimeError("MaxPool3d on the HIP path: (1, KH, KW) windows with stride (1, SH, SW), 'same' / 'valid' padding (the reference's stem pool, nnet/networks.py:470)")
    if mp.channels_last:
        x = x.movedim(-1, 1)
    same = mp.padding_type == "same"
    pad0 = (k[1] // 2, k[2] // 2) if same else (0, 0)
    pad1 = ((k[1] - 1) // 2, (k[2] - 1) // 2) if same else (0, 0)
    if x.shape[1] % 4:
        raise RuntimeError("MaxPool3d on the HIP path needs a channel count that is a multiple of 4")
    y = MaxPoolHWFn.apply(x, (k[1], k[2]), (s[1], s[2]), pad0, pad1)
    return y.movedim(1, -1) if mp.channels_last else y


class _RowsPoolFn(torch.autograd.Function):
    """non-overlapping average pooling (kernel = stride = P, T % P == 0) / nearest up-sampling by P of rows [B][T][D]"""

    @staticmethod
    def forward(ctx, x, P, up):
        rt.require_gpu(x)
        B, T, D = x.shape
        xa = x.to(rt.act_dtype()).contiguous()
        ctx.saved = (B, T, D, P, up, x.dtype)
        if up:
            y = torch.empty((B, T * P, D), dtype=xa.dtype, device=x.device)
            lib.upsample_rows(rt.dt(), xa.data_ptr(), y.data_ptr(), B, T, D, P, 0, rt.stream())
        else:
            y = torch.empty((B, T // P, D), dtype=xa.dtype, device=x.device)
            lib.patch_pool_fwd(rt.dt(), xa.data_ptr(), y.data_ptr(), B, T, D, P, rt.stream())
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        B, T, D, P, up, xdt = ctx.saved
        dya = dy.to(rt.act_dtype()).contiguous()
        dx = torch.empty((B, T, D), dtype=dya.dtype, device=dy.device)
        if up:
            lib.upsample_rows(rt.dt(), dya.data_ptr(), dx.data_ptr(), B, T, D, P, 1, rt.stream())
        else:
            lib.patch_pool_bwd(rt.dt(), dya.data_ptr(), dx.data_ptr(), B, T, D, P, rt.stream())
        return dx.to(xdt), None, None


def avgpool1d_module_forward(ap, x):
    k = ap.kernel_size[0] if isinstance(ap.kernel_size, tuple) else ap.kernel_size
    s = ap.stride[0] if isinstance(ap.stride, tuple) else ap.stride
    p = ap.padding[0] if isinstance(ap.padding, tuple) else ap.padding
    if not ap.channels_last:
        x = x.transpose(1, 2)
    if k != s or p != 0 or ap.ceil_mode or x.shape[2] % 4 or x.shape[1] < k:
        raise RuntimeError("AvgPool1d on the HIP path: kernel_size == stride, no padding, floor mode (the patch pooling of nnet/attentions.py:342-346), D % 4 == 0")
    y = _RowsPoolFn.apply(x[:, :x.shape[1] // k * k], k, False)                      # floor mode drops the ragged tail
    return y if ap.channels_last else y.transpose(1, 2)


def upsample_module_forward(up, x):
    sf = up.scale_factor[0] if isinstance(up.scale_factor, (tuple, list)) else up.scale_factor
    if up.mode != "nearest" or up.size is not None or sf is None or int(sf) != sf or x.dim() != 3:
        raise RuntimeError("Upsample on the HIP path: mode='nearest' with an integer scale_factor on a 3-D tensor (nnet/attentions.py:365-380)")
    if not up.channels_last:
        x = x.transpose(1, 2)
    if x.shape[2] % 4:
        raise RuntimeError("Upsample on the HIP path needs a feature width that is a multiple of 4")
    y = _RowsPoolFn.apply(x, int(sf), True)
    return y if up.channels_last else y.transpose(1, 2)


def global_avgpool2d_module_forward(gp, x, mask=None):
    if mask is not None or tuple(gp.dim) != (2, 3) or x.dim() != 4:
        raise RuntimeError("GlobalAvgPool2d on the HIP path: mean over dims (2, 3) of an (N, C, H, W) tensor without a mask")
    y = ops.AvgPoolFn.apply(x.permute(0, 2, 3, 1).to(rt.act_dtype()).contiguous()).to(x.dtype)
    return y[:, :, None, None] if gp.keepdim else y


class DepthwiseConv1dFn(torch.autograd.Function):
    """depthwise layers.Conv1d (groups == channels) on rows [B][T][C]: the fused GLU + depthwise-conv kernels of the ConvolutionModule with the gate pinned to 1
    (GLU(a, b) = a * sigmoid(b); sigmoid(40) rounds to exactly 1.0f): no separate kernel to keep in parity"""

    @staticmethod
    def forward(ctx, x, _anchor, conv):
        rt.require_gpu(x)
        B, T, C = x.shape
        K, stride, pl = conv.kernel_size[0], conv.stride[0], ops.dw_pad_left(conv)
        u = torch.cat([x.to(rt.act_dtype()), torch.full((B, T, C), 40.0, dtype=rt.act_dtype(), device=x.device)], dim=2).contiguous()
        To = (T - 1) // stride + 1
        out = torch.empty((B, To, C), dtype=rt.act_dtype(), device=x.device)
        lib.glu_dwconv_fwd(rt.dt(), u.data_ptr(), conv.weight.data_ptr(), ops._p(conv.bias), out.data_ptr(), None, B, T, C, K, stride, pl, rt.stream())
        ctx.saved = (u, conv, B, T, C, K, stride, pl, x.dtype)
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        u, conv, B, T, C, K, stride, pl, xdt = ctx.saved
        dya = dy.to(rt.act_dtype()).contiguous()
        du = torch.empty((B * T, 2 * C), dtype=dya.dtype, device=dy.device)
        lib.dwconv_glu_bwd(rt.dt(), dya.data_ptr(), u.data_ptr(), conv.weight.data_ptr(), du.data_ptr(), ops.grad_of(conv.weight).data_ptr(),
                           None if conv.bias is None else ops.grad_of(conv.bias).data_ptr(), B, T, C, K, stride, pl, rt.stream())
        return du.view(B, T, 2 * C)[:, :, :C].to(xdt), None, None


class Conv3dStemFn(torch.autograd.Function):
    """layers.Conv3d in the stem shape (Cin = 1, kernel (5,7,7), stride (1,2,2), 'same') on a logical (B, 1, T, H, W) tensor: im2col + GEMM (nnet/layers.py:326-503);
    no input gradient (the clip)"""

    @staticmethod
    def forward(ctx, x, _anchor, conv):
        rt.require_gpu(x)
        B, _, T, H, W = x.shape
        v = x.reshape(B, T, H, W).float().contiguous()
        C = conv.weight.shape[0]
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        M = B * T * OH * OW
        sh = rt.shadow(conv.weight)
        K, Kp = sh.Tm * sh.C, sh.Cp
        A = ops.empty((M, Kp), rt.act_dtype(), v)
        lib.stem_im2col(rt.dt(), v.data_ptr(), A.data_ptr(), B, T, H, W, Kp, rt.stream())
        y = ops.empty((M, C), rt.act_dtype(), v)
        ops.gemm_nt(A, sh.fwd, y, M, C, Kp, bias=conv.bias)
        ctx.saved = (A, conv, M, C, K)
        return y.view(B, T, OH, OW, C).permute(0, 4, 1, 2, 3).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        A, conv, M, C, K = ctx.saved
        dya = dy.permute(0, 2, 3, 4, 1).to(rt.act_dtype()).contiguous().view(M, C)
        if conv.bias is not None:
            ops.colsum(dya, C, ops.grad_of(conv.bias), M, C)
        ops.gemm_tn(dya, A, ops.grad_of(conv.weight), M, C, K, q_rows=ops.rows_plain(A.shape[1]))
        return None, None, None


def conv3d_module_forward(conv, x):
    stem = conv.in_channels == 1 and tuple(conv.kernel_size) == (5, 7, 7) and tuple(conv.stride) == (1, 2, 2) and conv.padding_type == "same" and conv.mask is None
    if not stem or not hasattr(conv.weight, "_avec_shadow"):
        raise RuntimeError("Conv3d on the HIP path: the Cin = 1, (5,7,7) / (1,2,2) 'same' stem shape (nnet/networks.py:459-468)")
    if conv.channels_last:
        x = x.movedim(-1, 1)
    y = Conv3dStemFn.apply(x, conv.weight, conv)
    return y.movedim(1, -1) if conv.channels_last else y
