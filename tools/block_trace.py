"""Per-launch timeline of ONE conformer block (forward + backward) INSIDE a captured graph, without a profiler: every C-ABI call of the block is followed by a
one-wave stamp kernel (avec_stamp) that writes the wall clock; the stamps replay with the graph.  Prints, per call, the time since the previous stamp minus the
cost of the stamp node itself (measured on a chain of stamps: tools/probes/graph_chain.hip says 1.6 us per dependent node).
    python tools/block_trace.py [B T D]        (one MI355X)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avec_amd  # noqa: E402
import nnet  # noqa: E402
from avec_amd import ops, runtime as rt  # noqa: E402
from avec_amd import lib as libmod  # noqa: E402

TR = {"on": False, "names": [], "buf": None}
SKIP = {"stamp", "version", "last_error", "struct_size", "set_reduce_workspace", "set_zero_page"}


def install():
    L = libmod.lib
    L.load()
    raw_stamp = L.raw("avec_stamp")
    for k in [k for k in L.__dict__ if k != "_dll"]:
        del L.__dict__[k]
    orig = libmod._Lib.__getattr__

    def traced(self, name):
        call = orig(self, name)
        if name in SKIP:
            return call

        def wrapped(*args):
            call(*args)
            if TR["on"] and len(TR["names"]) < 1000:
                kn = L.raw("avec_last_kernel")() if name.startswith("gemm") else None
                TR["names"].append(name + (" [" + kn.decode() + "]" if kn else ""))
                raw_stamp(TR["buf"].data_ptr(), len(TR["names"]), rt.stream())
        wrapped.__name__ = name
        self.__dict__[name] = wrapped
        return wrapped
    libmod._Lib.__getattr__ = traced


def main():
    B, T, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 100, 256)
    avec_amd.set_compute_dtype("bf16")
    dev = torch.device("cuda:0")
    install()
    L = libmod.lib
    TR["buf"] = torch.zeros(1024, dtype=torch.int64, device=dev)
    att = {"class": "RelPos1dMultiHeadAttention", "params": dict(num_heads=4, attn_drop_rate=0.0, num_pos_embeddings=10000, weight_init="default", bias_init="default")}
    blk = nnet.ConformerBlock(dim_model=D, dim_expand=D, ff_ratio=4, drop_rate=0.1, att_params=att, conv_stride=1,
                              conv_params={"class": "Conv1d", "params": {"padding": "same", "kernel_size": 15}}).to(dev).train()
    x = torch.randn(B, T, D, device=dev)
    wgt = torch.randn(B, T, D, device=dev)
    lens = torch.full((B,), T, dtype=torch.int64, device=dev)
    from avec_amd.nnet.modules import LengthMask
    mask = LengthMask(lens)
    xg = x.clone().requires_grad_(True)

    def fwdbwd():
        rt.reset_zero_pool(dev)
        L.raw("avec_stamp")(TR["buf"].data_ptr(), 0, rt.stream())
        y = blk(xg, mask=mask)
        y.backward(wgt)
        ops.flush_param_grads(all_streams=True)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fwdbwd()
    torch.cuda.synchronize()
    # cost of a stamp node: a chain of 100 stamps
    gs = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gs):
        for i in range(100):
            L.raw("avec_stamp")(TR["buf"].data_ptr(), 1000 + (i & 1), rt.stream())
    for _ in range(3):
        gs.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gs.replay()
    e1.record()
    torch.cuda.synchronize()
    stamp_us = e0.elapsed_time(e1) * 1e3 / 1000
    g = torch.cuda.CUDAGraph()
    TR["on"], TR["names"] = True, []
    with torch.cuda.graph(g):
        fwdbwd()
    TR["on"] = False
    n = len(TR["names"])
    acc = torch.zeros(n + 1, dtype=torch.float64)
    R = 20
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for _ in range(R):
        g.replay()
        torch.cuda.synchronize()
        t = TR["buf"][:n + 1].cpu().double()
        acc += (t - t[0]) / 100.0
    acc /= R
    print("B=%d T=%d D=%d: %d launches, stamp node %.2f us (subtracted), block forward + backward %.1f us with stamps, %.1f us net" % (B, T, D, n, stamp_us, acc[n], acc[n] - n * stamp_us))
    for i in range(n):
        print("%4d %8.1f  %6.2f  %s" % (i, acc[i + 1], acc[i + 1] - acc[i] - stamp_us, TR["names"][i]))


if __name__ == "__main__":
    main()
