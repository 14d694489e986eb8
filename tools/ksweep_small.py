"""Fixed vs per-K cost of the small conformer products (64x64 tiles): M = 3200 / 1600, N = 256 / 1024 / 360, K sweep; timed inside a hipGraph of 50 back-to-back dependent launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import ops
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda"); adt = torch.bfloat16
def t(M, N, K, act=0, res=False):
    A = torch.randn(M, K, device=d).to(adt); W = torch.randn(N, K, device=d).to(adt)
    bias = torch.randn(N, device=d)
    out = torch.empty(M, N, device=d, dtype=torch.float32 if res else adt)
    r = torch.randn(M, N, device=d) if res else None
    def run():
        for _ in range(50):
            ops.gemm_nt(A, W, out, M, N, K, bias=bias, act=act, res=r, out_f32=res)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 500 * 1e3
    print("M %5d N %5d K %5d act %d res %d : %6.2f us  %6.1f TF" % (M, N, K, act, res, us, 2.0 * M * N * K / us / 1e6), flush=True)
for (M, N) in ((3200, 256), (3200, 1024), (1600, 360), (800, 1440)):
    for K in (64, 128, 256, 512, 1024, 1440):
        t(M, N, K)
t(3200, 256, 1024, res=True); t(3200, 1024, 256, act=1)
