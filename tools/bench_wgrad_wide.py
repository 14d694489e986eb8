"""Grouped 3x3 weight gradients of ResNet stages 2-4 (avec_wgrad3x3_c128_grouped: 9 layers in one launch, B = 32 clips of 100 frames = 3200 images) and the single
layers, timed inside a captured graph.  AVEC_NO_WGRAD_PAIRS=1 selects the slab kernel of round 3.   usage: PYTHONPATH=. python tools/bench_wgrad_wide.py"""
import torch
import avec_amd
from avec_amd import runtime as rt
from avec_amd.lib import WgradItem, lib

import os
NIMG = int(os.environ.get("WG_IMAGES", "3200"))
GEOS = [(NIMG, 128, 11, 11), (NIMG, 256, 6, 6), (NIMG, 512, 3, 3)]


def timed(fn, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (4 * reps)


def main():
    d = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    items, keep, flops = [], [], 0.0
    for N, C, H, W in GEOS:
        x = torch.randn(N, H, W, C, device=d).to(torch.bfloat16)
        dy = torch.randn(N * H * W, C, device=d).to(torch.bfloat16)
        fl = 2.0 * N * H * W * C * 9 * C
        dw1 = torch.zeros(C, 9 * C, device=d)
        us = timed(lambda: lib.wgrad3x3_c128(x.data_ptr(), dy.data_ptr(), dw1.data_ptr(), N, C, H, W, rt.stream()))
        print("single  %4d ch %2dx%-2d  %8.1f us  %7.1f TFLOP/s (nominal)   %s" % (C, H, W, us, fl / us / 1e6, lib.raw("avec_last_kernel")().decode()))
        for _ in range(3):
            dw = torch.zeros(C, 9 * C, device=d)
            it = WgradItem()
            it.x, it.dy, it.dw, it.images, it.C, it.H, it.W = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, C, H, W
            items.append(it); keep.append((x, dy, dw)); flops += fl
    arr = (WgradItem * len(items))(*items)
    us = timed(lambda: lib.wgrad3x3_c128_grouped(arr, len(items), rt.stream()))
    print("grouped 9 layers          %8.1f us  %7.1f TFLOP/s (nominal)   %s" % (us, flops / us / 1e6, lib.raw("avec_last_kernel")().decode()))


if __name__ == "__main__":
    main()
