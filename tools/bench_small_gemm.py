"""Latency of the conformer-sized NT products (avec_gemm_nt, bf16) measured inside a captured graph of REPS back-to-back launches (no host overhead):
us per launch, TFLOP/s, and the kernel instance the dispatcher chose.  usage: python tools/bench_small_gemm.py [reps]"""
import sys
import torch
import avec_amd
from avec_amd import ops, runtime as rt
from avec_amd.lib import lib, ACT_SWISH, ACT_NONE

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
SHAPES = [  # (M, N, K, kind)   kind: plain | ffn1 (bias + swish) | res (bias + fp32 residual out)
    (1600, 1440, 360, "ffn1"), (1600, 360, 1440, "res"), (1600, 1080, 360, "plain"), (1600, 360, 360, "res"),
    (3200, 1024, 256, "ffn1"), (3200, 256, 1024, "res"), (3200, 768, 256, "plain"), (3200, 256, 256, "res"),
    (6400, 720, 180, "ffn1"), (6400, 180, 720, "res"), (6400, 540, 180, "plain"), (6400, 180, 180, "res"),
    (800, 1440, 360, "ffn1"), (800, 360, 1440, "res"), (800, 1080, 360, "plain"), (800, 360, 360, "res"),
    (199, 360, 360, "plain"), (399, 180, 180, "plain"),
]


def main():
    dev = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(0)
    print("%-24s %-6s %8s %9s  %s" % ("M x N x K", "kind", "us", "TFLOP/s", "kernel"))
    for M, N, K, kind in SHAPES:
        A = torch.randn(M, K, generator=g).bfloat16().to(dev)
        W = (0.05 * torch.randn(N, K, generator=g)).bfloat16().to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(dev)
        out = torch.empty(M, N, dtype=torch.float32 if kind == "res" else torch.bfloat16, device=dev)
        kw = dict(bias=bias)
        if kind == "ffn1":
            kw.update(act=ACT_SWISH)
        if kind == "res":
            kw.update(res=res, alpha=0.5, out_f32=True)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                ops.gemm_nt(A, W, out, M, N, K, **kw)
            name = lib.raw("avec_last_kernel")()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(REPS):
                    ops.gemm_nt(A, W, out, M, N, K, **kw)
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / (5 * REPS)
        print("%-24s %-6s %8.2f %9.1f  %s" % ("%d x %d x %d" % (M, N, K), kind, us, 2.0 * M * N * K / us / 1e6, name if isinstance(name, str) else name.decode()))


if __name__ == "__main__":
    main()
