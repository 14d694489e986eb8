"""Ablation of the NT GEMM kernel (tools/build_abl.sh variants): time of the same launch with pieces of the inner loop removed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
    print("%-40s %8.1f us  %8.1f TFLOP/s-equivalent" % (name, ms * 1e3, flops / ms / 1e9), flush=True)

def plain(M, N, K):
    A = torch.randn(M, K, device=d).to(adt); W = torch.randn(N, K, device=d).to(adt); out = torch.empty(M, N, device=d, dtype=adt)
    timeit(lambda: ops.gemm_nt(A, W, out, M, N, K), 2.0 * M * N * K, "nt plain %dx%dx%d" % (M, N, K))

def conv(Nimg, H, Cin, Cout):
    x = torch.randn(Nimg, H, H, Cin, device=d).to(adt); M = Nimg * H * H
    W = torch.randn(Cout, 9 * Cin, device=d).to(adt); y = torch.empty(M, Cout, device=d, dtype=adt)
    rows = ops.rows_conv(H, H, Cin, 3, 3, 1, 1, H, H)
    timeit(lambda: ops.gemm_nt(x, W, y, M, Cout, 9 * Cin, rows=rows, mode=ROWS_CONV_FWD), 2.0 * M * Cout * 9 * Cin, "conv fwd %dx%d^2 %d->%d" % (Nimg, H, Cin, Cout))

print("lib:", os.environ.get("AVEC_LIB_PATH", "default"), "RB", os.environ.get("AVEC_NT_RB"), "STG", os.environ.get("AVEC_NT_STG"))
plain(4096, 4096, 4096)
conv(3200, 11, 128, 128)
conv(3200, 6, 256, 256)
conv(3200, 3, 512, 512)
