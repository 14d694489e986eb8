"""where a chain-kernel workgroup spends its time: s_memtime stamps of the first and the last workgroup (avec_chain_debug_stamps)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avec_amd  # noqa: E402
import nnet  # noqa: E402
from avec_amd import ops, runtime as rt  # noqa: E402
from avec_amd.lib import lib  # noqa: E402

avec_amd.set_compute_dtype("bf16")
dev = torch.device("cuda:0")
NAMES = ["operand", "productA", "middle", "copies", "productB", "epilogue"]
dbg = torch.zeros(16, dtype=torch.int64, device=dev)
for (M, D, F) in [(3200, 256, 1024), (1600, 360, 1440), (100, 256, 1024)]:
    mod = nnet.FeedForwardModule(D, F, 0.1, "Swish", True).to(dev).train()
    x = torch.randn(1, M, D, device=dev, requires_grad=True)
    wgt = torch.randn(1, M, D, device=dev)
    for it in range(3):
        rt.reset_zero_pool(dev)
        lib.raw("avec_chain_debug_stamps")(dbg.data_ptr())
        y = mod.residual_forward(x, 0.5)
        torch.cuda.synchronize()
        f = dbg.cpu().tolist()
        y.backward(wgt)
        torch.cuda.synchronize()
        b = dbg.cpu().tolist()
        lib.raw("avec_chain_debug_stamps")(None)
    for tag, t in (("fwd", f), ("bwd", b)):
        for w, o in (("first wg", 0), ("last wg", 8)):
            d = [t[o + i + 1] - t[o + i] for i in range(6)]
            print("M=%d D=%d %s %s: " % (M, D, tag, w) + "  ".join("%s %d" % (n, v) for n, v in zip(NAMES, d)) + "  | total %d  start-skew %d" % (t[o + 6] - t[o], t[o] - t[0]), flush=True)
