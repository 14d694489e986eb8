"""The four 64-channel 3x3 weight gradients of ResNet stage 1 (avec_wgrad3x3_c64_grouped: 3200 images of 22x22) timed inside a captured graph.
usage: PYTHONPATH=. [AVEC_LIB_PATH=tools/_bin/libavec_c3wabl_N.so] python tools/bench_wgrad_c64.py"""
import os
import torch
import avec_amd
from avec_amd import runtime as rt
from avec_amd.lib import WgradItem, lib
from bench_wgrad_wide import timed

NIMG = int(os.environ.get("WG_IMAGES", "3200"))


def main():
    d = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    items, keep, flops = [], [], 0.0
    N, C, H, W = NIMG, 64, 22, 22
    for _ in range(4):
        x = torch.randn(N, H, W, C, device=d).to(torch.bfloat16)
        dy = torch.randn(N * H * W, C, device=d).to(torch.bfloat16)
        dw = torch.zeros(C, 9 * C, device=d)
        it = WgradItem()
        it.x, it.dy, it.dw, it.images, it.C, it.H, it.W = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, C, H, W
        items.append(it); keep.append((x, dy, dw)); flops += 2.0 * N * H * W * C * 9 * C
    arr = (WgradItem * len(items))(*items)
    us = timed(lambda: lib.wgrad3x3_c64_grouped(arr, len(items), rt.stream()))
    byt = 4 * 2 * N * H * W * C * 2
    print("%s grouped 4 layers  %8.1f us  %7.1f TFLOP/s  %6.2f TB/s (algorithmic)   %s" % (os.environ.get("AVEC_LIB_PATH", "default"), us, flops / us / 1e6, byt / us / 1e6, lib.raw("avec_last_kernel")().decode()))
    try:
        import ctypes, numpy as np
        f = lib.raw("avec_c3w_debug")
        out = (ctypes.c_float * 128)()
        f(out)
        v = np.array(out[:]).reshape(16, 8).mean(0)
        print("  in-kernel cycles of wave 0 (mean of 16 workgroups): waits before G0 %.0f  G1 %.0f  G2 %.0f  G3 %.0f   k-step groups %.0f" % (v[0], v[1], v[2], v[3], v[4]))
    except Exception:
        pass
    if not os.environ.get("AVEC_LIB_PATH"):
        xf, dyf = keep[0][0].float(), keep[0][1].float().view(N, H, W, C)
        ref = torch.nn.grad.conv2d_weight(xf[:64].permute(0, 3, 1, 2), (C, C, 3, 3), dyf[:64].permute(0, 3, 1, 2), padding=1)      # [co][ci][kh][kw]
        dw = torch.zeros(C, 9 * C, device=d)
        lib.wgrad3x3_c64(keep[0][0].data_ptr(), keep[0][1].data_ptr(), dw.data_ptr(), 64, H, W, rt.stream())
        got = dw.view(C, 3, 3, C).permute(0, 3, 1, 2)
        print("check vs torch (64 images): rel err %.2e" % ((got - ref).norm() / ref.norm()).item())


if __name__ == "__main__":
    main()
