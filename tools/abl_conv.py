"""Shifted-window convolution kernel on the ResNet stage 2-4 shapes: forward (+ BatchNorm statistics) and backward-data (+ residual gradient), us per launch, and a
channel sweep (K = 9 Cin) at fixed tile count for the per-tile fixed cost (time = a + b K).  AVEC_LIB_PATH selects an ablation build (tools/build_abl.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import ops, runtime as rt
from avec_amd.lib import lib, ROWS_CONV_FWD, ROWS_CONV_BWD
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda"); adt = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def conv(Nimg, H, Cin, Cout, tag=""):
    x = torch.randn(Nimg, H, H, Cin, device=d).to(adt); M = Nimg * H * H
    W = (0.05 * torch.randn(Cout, 9 * Cin, device=d)).to(adt); Wb = (0.05 * torch.randn(Cin, 9 * Cout, device=d)).to(adt)
    y = torch.empty(M, Cout, device=d, dtype=adt); st = torch.zeros(64 * 2 * Cout, device=d)
    dx = torch.empty(M, Cin, device=d, dtype=adt); res = torch.randn(M, Cin, device=d).to(adt)
    rows = ops.rows_conv(H, H, Cin, 3, 3, 1, 1, H, H); rb = ops.rows_conv(H, H, Cout, 3, 3, 1, 1, H, H)
    fl = 2.0 * M * Cout * 9 * Cin
    tf = timeit(lambda: ops.gemm_nt(x, W, y, M, Cout, 9 * Cin, rows=rows, mode=ROWS_CONV_FWD, stats=st))
    name = lib.raw("avec_last_kernel")(); name = name if isinstance(name, str) else name.decode()
    tb = timeit(lambda: ops.gemm_nt(y, Wb, dx, M, Cin, 9 * Cout, rows=rb, mode=ROWS_CONV_BWD, res=res, res_act=True))
    print("%s conv %dx%d^2 %4d->%4d  fwd+stats %7.1f us %7.1f TF | bwd+res %7.1f us %7.1f TF  %s" % (tag, Nimg, H, Cin, Cout, tf, fl / tf / 1e6, tb, fl / tb / 1e6, name[:40]), flush=True)


print("lib:", os.environ.get("AVEC_LIB_PATH", "default"))
conv(3200, 11, 128, 128)
conv(3200, 6, 256, 256)
conv(3200, 3, 512, 512)
if "--sweep" in sys.argv:
    for c in (32, 64, 128, 256): conv(3200, 11, c, 128, "sweep")
    for c in (64, 128, 256, 512): conv(3200, 6, c, 256, "sweep")
    for c in (128, 256, 512, 1024): conv(3200, 3, c, 512, "sweep")
