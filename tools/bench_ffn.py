"""Micro-benchmark of the fused FFN module against the unfused launch sequence (one MI355X): python tools/bench_ffn.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avec_amd
import nnet
from avec_amd import ops


def main():
    avec_amd.set_compute_dtype("bf16")
    dev = torch.device("cuda:0")
    for M, D in [(6400, 256), (3200, 256), (1600, 360), (3200, 360)]:
        mod = nnet.FeedForwardModule(D, 4 * D, 0.1, "Swish", True).to(dev).train()
        x = torch.randn(M // 50, 50, D, device=dev)
        w = torch.randn_like(x)
        for fused in (False, True):
            ops.FFN_FUSED = fused
            for phase in ("fwd", "fwd+bwd"):
                def run():
                    xg = x.clone().requires_grad_(phase != "fwd")
                    y = mod.residual_forward(xg, 0.5)
                    if phase != "fwd":
                        (y * w).sum().backward()
                for _ in range(5):
                    run()
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    run()
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g):
                        for _ in range(10):
                            run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 200 * 1e6
                fl = 4.0 * M * D * 4 * D * (1 if phase == "fwd" else 3)
                print("M=%d D=%d %-8s fused=%d  %.1f us  %.0f TFLOP/s" % (M, D, phase, fused, us, fl / us / 1e6))


if __name__ == "__main__":
    main()
