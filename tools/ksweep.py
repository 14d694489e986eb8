"""time = a + b*K fit of the NT GEMM at the ResNet stage-2 tile count (M = 387200 rows, N = 128): fixed per-tile cost vs per-K-step cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import ops
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda"); adt = torch.bfloat16
def t(M, N, K, iters=20):
    A = torch.randn(M, K, device=d).to(adt); W = torch.randn(N, K, device=d).to(adt); out = torch.empty(M, N, device=d, dtype=adt)
    for _ in range(3): ops.gemm_nt(A, W, out, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_nt(A, W, out, M, N, K)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print("M %d N %d K %5d : %7.1f us  %7.1f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
print("RB", os.environ.get("AVEC_NT_RB"))
for N in (128, 256):
    for K in (64, 128, 256, 512, 1152, 2304):
        t(387200 if N == 128 else 115200, N, K)
