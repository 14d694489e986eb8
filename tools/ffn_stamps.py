import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, avec_amd, nnet
from avec_amd import ops
from avec_amd.lib import lib
avec_amd.set_compute_dtype("bf16")
dev = torch.device("cuda:0")
for M, D in [(3200, 256), (1600, 360)]:
    mod = nnet.FeedForwardModule(D, 4 * D, 0.1, "Swish", True).to(dev).train()
    x = torch.randn(M // 50, 50, D, device=dev)
    for it in range(3):
        xg = x.clone().requires_grad_(True)
        y = mod.residual_forward(xg, 0.5)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 8)()
        lib.raw("avec_ffn_debug_stamps")(buf)
        f = list(buf)
        y.sum().backward()
        torch.cuda.synchronize()
        lib.raw("avec_ffn_debug_stamps")(buf)
        b = list(buf)
    print("M=%d D=%d fwd cycles: prologue %d loop %d epilogue %d | bwd: prologue %d loop %d epilogue %d (100 MHz ticks x?)" % (M, D, f[1]-f[0], f[2]-f[1], f[3]-f[2], b[1]-b[0], b[2]-b[1], b[3]-b[2]))
