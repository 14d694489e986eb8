"""Visual stem forward + backward alone (B = 32 clips of 100 x 88 x 88), per mode: stem3p (pool inside the convolution kernel, recompute in backward), stem3d (round-1 direct
kernels), with HIP events.      python tools/bench_stem.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import avec_amd
    import nnet
    from avec_amd import ops
    dev = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    torch.manual_seed(0)
    B, T, H, W = 32, 100, 88, 88
    x = torch.randn(B, T, H, W, device=dev)
    stem = nnet.ConvNeuralNetwork(dim_input=1, dim_layers=64, kernel_size=(5, 7, 7), strides=(1, 2, 2), norm="BatchNorm3d", act_fun="ReLU", drop_rate=0.0, dim=3).to(dev).train()
    conv, bn = stem.layers[0][0], stem.layers[0][1]
    wout = None
    for name, p3 in (("stem3p", True), ("stem3d", False), ("stem3p", True), ("stem3d", False)):
        ops.STEM3P_FUSED_WGRAD = os.environ.get("AVEC_STEM3P_FUSED", "1") != "0"
        ops.STEM3P = p3
        ts = []
        for it in range(6):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            y = ops.VideoStemFn.apply(x, conv.weight, conv, bn, True)
            e[1].record()
            if wout is None:
                wout = torch.randn(y.shape, device=dev).to(torch.bfloat16)
            y.backward(wout)
            e[2].record()
            torch.cuda.synchronize()
            ts.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
        f = min(t[0] for t in ts[2:]); b = min(t[1] for t in ts[2:])
        print("%-7s forward %.3f ms  backward %.3f ms  total %.3f ms" % (name, f, b, f + b), flush=True)


if __name__ == "__main__":
    main()
