"""Stride-2 3x3 / 1x1 convolutions of the ResNet stage boundaries in isolation: forward (+ statistics), backward-data (+ residual), weight gradient; us and TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import ops, runtime as rt
from avec_amd.lib import lib, ROWS_CONV_FWD, ROWS_CONV_BWD
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda"); adt = torch.bfloat16
last = lambda: (lambda n: n if isinstance(n, str) else n.decode())(lib.raw("avec_last_kernel")())


def timeit(fn, flops, name, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-34s %8.1f us  %7.1f TFLOP/s  %s" % (name, ms * 1e3, flops / ms / 1e9, last()))


def conv(Nimg, H, Cin, Cout, k):
    pad = (k - 1) // 2
    x = torch.randn(Nimg, H, H, Cin, device=d).to(adt)
    OH = (H - 1) // 2 + 1
    M = Nimg * OH * OH
    W = torch.randn(Cout, k * k * Cin, device=d).to(adt)
    Wb = torch.randn(Cin, k * k * Cout, device=d).to(adt)
    y = torch.empty(M, Cout, device=d, dtype=adt)
    st = torch.zeros(64 * 2 * Cout, device=d)
    rows = ops.rows_conv(H, H, Cin, k, k, 2, pad, OH, OH)
    fl = 2.0 * M * Cout * k * k * Cin
    tag = "%dx%d^2 %d->%d k%d" % (Nimg, H, Cin, Cout, k)
    timeit(lambda: ops.gemm_nt(x, W, y, M, Cout, k * k * Cin, rows=rows, mode=ROWS_CONV_FWD, stats=st), fl, "fwd   " + tag)
    dx = torch.empty(Nimg * H * H, Cin, device=d, dtype=adt)
    res = torch.randn(Nimg * H * H, Cin, device=d).to(adt)
    rb = ops.rows_conv(H, H, Cout, k, k, 2, pad, OH, OH)
    if k == 3:
        timeit(lambda: ops.gemm_nt(y, Wb, dx, Nimg * H * H, Cin, k * k * Cout, rows=rb, mode=ROWS_CONV_BWD, res=res, res_act=True), fl, "bwd   " + tag)
    dW = torch.zeros(Cout, k * k * Cin, device=d)
    timeit(lambda: ops.gemm_tn(y, x, dW, M, Cout, k * k * Cin, q_rows=rows, q_mode=ROWS_CONV_FWD), fl, "wgrad " + tag)


for k in (3, 1):
    conv(3200, 22, 64, 128, k)
    conv(3200, 11, 128, 256, k)
    conv(3200, 6, 256, 512, k)
