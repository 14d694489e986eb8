"""Experiment: wide tiles (AVEC_NT_WIDE=1: 128x256, 2: 256x256, 3: 256x128) of the implicit-GEMM kernel: correctness vs torch + time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import avec_amd
from avec_amd import ops, runtime as rt
from avec_amd.lib import ROWS_CONV_FWD, ROWS_CONV_BWD
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda"); adt = torch.bfloat16

def timeit(fn, flops, name, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return "%-34s %7.1f us %7.1f TF" % (name, ms * 1e3, flops / ms / 1e9)

def conv(Nimg, H, Cin, Cout, stride=1):
    torch.manual_seed(0)
    x = torch.randn(Nimg, H, H, Cin, device=d).to(adt)
    OH = (H - 1) // stride + 1; M = Nimg * OH * OH
    Wt = (torch.randn(Cout, Cin, 3, 3, device=d) / (3 * Cin ** 0.5))
    W = Wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(adt).contiguous()
    Wb = Wt.permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).to(adt).contiguous()       # [Cin][kh][kw][Cout]
    y = torch.empty(M, Cout, device=d, dtype=adt)
    st = torch.zeros(64 * 2 * Cout, device=d)
    rows = ops.rows_conv(H, H, Cin, 3, 3, stride, 1, OH, OH)
    ops.gemm_nt(x, W, y, M, Cout, 9 * Cin, rows=rows, mode=ROWS_CONV_FWD, stats=st)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), W.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), stride=stride, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    e_f = ((y.float() - ref).norm() / ref.norm()).item()
    s = st.view(64, 2, Cout).sum(0)
    e_s = ((s[0] - ref.sum(0)).norm() / ref.sum(0).norm()).item(); e_q = ((s[1] - (ref * ref).sum(0)).norm() / (ref * ref).sum(0).norm()).item()
    fl = 2.0 * M * Cout * 9 * Cin
    t1 = timeit(lambda: ops.gemm_nt(x, W, y, M, Cout, 9 * Cin, rows=rows, mode=ROWS_CONV_FWD, stats=st), fl, "fwd+stats")
    dy = torch.randn(M, Cout, device=d).to(adt)
    dx = torch.empty(Nimg * H * H, Cin, device=d, dtype=adt)
    rb = ops.rows_conv(H, H, Cout, 3, 3, stride, 1, OH, OH)
    ops.gemm_nt(dy, Wb, dx, Nimg * H * H, Cin, 9 * Cout, rows=rb, mode=ROWS_CONV_BWD)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, W.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), stride=stride, padding=1)
    yr.backward(dy.float().reshape(Nimg, OH, OH, Cout).permute(0, 3, 1, 2))
    dref = xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin)
    e_b = ((dx.float() - dref).norm() / dref.norm()).item()
    t2 = timeit(lambda: ops.gemm_nt(dy, Wb, dx, Nimg * H * H, Cin, 9 * Cout, rows=rb, mode=ROWS_CONV_BWD), fl, "bwd-data")
    print("%dx%d^2 %d->%d s%d | err fwd %.1e stats %.1e %.1e bwd %.1e | %s | %s" % (Nimg, H, Cin, Cout, stride, e_f, e_s, e_q, e_b, t1, t2), flush=True)

print("AVEC_NT_WIDE =", os.environ.get("AVEC_NT_WIDE"))
conv(3200, 11, 128, 128)
conv(3200, 6, 256, 256)
conv(3200, 3, 512, 512)
conv(3200, 22, 64, 128, 2)
conv(3200, 11, 128, 256, 2)
conv(3200, 6, 256, 512, 2)
conv(801, 6, 256, 256)
