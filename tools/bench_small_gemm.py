"""NT GEMM on the conformer shapes under the launcher's environment switches (tile size, LDS-DMA vs register staging): python tools/bench_small_gemm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import ops, runtime as rt

avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda")
SHAPES = [(3200, 1024, 256), (3200, 256, 1024), (3200, 768, 256), (3200, 256, 256), (3200, 512, 256), (1600, 1440, 360), (1600, 360, 1440), (1600, 1080, 360), (1600, 360, 360),
          (6400, 1024, 256), (6400, 256, 1024)]
tot = 0.0
for M, N, K in SHAPES:
    A = torch.randn(M, K, device=d).to(torch.bfloat16)
    W = torch.randn(N, K, device=d).to(torch.bfloat16)
    out = torch.empty(M, N, device=d, dtype=torch.bfloat16)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.gemm_nt(A, W, out, M, N, K)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops.gemm_nt(A, W, out, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    tot += us
    print("%5d x %4d x %4d  %6.1f us  %6.0f TFLOP/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
print("sum %.1f us" % tot)
