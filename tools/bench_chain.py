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
    # whole conformer blocks (the feed-forward kernels leave their output as partial sums for the next launch: the block is the fair unit)
    for (B, T, D) in [(32, 100, 256), (32, 50, 360), (1, 100, 256)]:
        att = {"class": "RelPos1dMultiHeadAttention", "params": dict(num_heads=4, attn_drop_rate=0.0, num_pos_embeddings=10000, weight_init="default", bias_init="default")}
        blk = nnet.ConformerBlock(dim_model=D, dim_expand=D, ff_ratio=4, drop_rate=0.1, att_params=att, conv_stride=1,
                                  conv_params={"class": "Conv1d", "params": {"padding": "same", "kernel_size": 15}}).to(dev).train()
        x = torch.randn(B, T, D, device=dev)
        wgt = torch.randn(B, T, D, device=dev)
        lens = torch.full((B,), T, dtype=torch.int64, device=dev)
        from avec_amd.nnet.modules import LengthMask
        mask = LengthMask(lens)
        for chain, lng in ((False, False), (False, True), (True, True)):
            ops.FFN_CHAIN, ops.LN_GEMM = chain, lng

            def fwd():
                rt.reset_zero_pool(dev)
                with torch.no_grad():
                    return blk(x, mask=mask)

            xg = x.clone().requires_grad_(True)

            def fwdbwd():
                rt.reset_zero_pool(dev)
                y = blk(xg, mask=mask)
                y.backward(wgt)
                ops.flush_param_grads(all_streams=True)
            tf, tfb = graphed(fwd, 5), graphed(fwdbwd, 5)
            print("block B=%2d T=%3d D=%3d ffn_chain=%d ln_gemm=%d: fwd %7.1f us   fwd+bwd(+wgrad) %7.1f us" % (B, T, D, chain, lng, tf, tfb), flush=True)


if __name__ == "__main__":
    main()
