"""Micro-benchmark of the row-resident chain kernels (csrc/chain.hip) against the launch sequences they replace, conformer shapes of the bench step.
    python tools/bench_chain.py        (one MI355X)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avec_amd  # noqa: E402
import nnet  # noqa: E402
from avec_amd import ops, runtime as rt  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def graphed(fn, reps=20):
    """the same call sequence replayed from a hipGraph (no host launch overhead): microseconds per call"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return timeit(g.replay, 20) / reps


def main():
    avec_amd.set_compute_dtype("bf16")
    dev = torch.device("cuda:0")
    for (M, D, F) in [(3200, 256, 1024), (1600, 360, 1440), (6400, 256, 1024), (800, 360, 1440), (100, 256, 1024)]:
        mod = nnet.FeedForwardModule(D, F, 0.1, "Swish", True).to(dev).train()
        x = torch.randn(1, M, D, device=dev)
        wgt = torch.randn(1, M, D, device=dev)
        for chain in (False, True):
            ops.FFN_CHAIN = chain

            def fwd():
                rt.reset_zero_pool(dev)
                with torch.no_grad():
                    return mod.residual_forward(x, 0.5)

            xg = x.clone().requires_grad_(True)

            def fwdbwd():
                rt.reset_zero_pool(dev)
                y = mod.residual_forward(xg, 0.5)
                y.backward(wgt)
                ops.flush_param_grads(all_streams=True)
            tf, tfb = graphed(fwd), graphed(fwdbwd, 10)
            print("FFN M=%5d D=%3d F=%4d chain=%d: fwd %7.1f us   fwd+bwd(+wgrad) %7.1f us" % (M, D, F, chain, tf, tfb), flush=True)
    for (M, D, N) in [(3200, 256, 768), (3200, 256, 512), (1600, 360, 1080), (1600, 360, 720)]:
        x = torch.randn(M, D, device=dev)
        lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        W = torch.randn(N, D, device=dev).bfloat16()
        bias = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t1 = graphed(lambda: ops.ln_gemm(x, lw, lb, 1e-6, W, D, bias, M, D, N))

        def two():
            h, _, _ = ops.layernorm_fwd(x, lw, lb, M, D, False, 1e-6)
            ops.gemm_nt(h, W, out, M, N, D, bias=bias)
        t2 = graphed(two)
        print("LN+GEMM M=%5d D=%3d N=%4d: one launch %6.1f us   two launches %6.1f us" % (M, D, N, t1, t2), flush=True)


if __name__ == "__main__":
    main()
