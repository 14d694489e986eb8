"""Stage-1 slab kernels in isolation (3200 images of 22x22x64): forward + statistics, backward-data + residual, weight gradient; us per launch, TFLOP/s, GB/s of tensors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import runtime as rt
from avec_amd.lib import lib

avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda")
N, H = int(os.environ.get("SLAB_N", "3200")), 22
M = N * H * H
x = torch.randn(N, H, H, 64, device=d).to(torch.bfloat16)
y = torch.randn(M, 64, device=d).to(torch.bfloat16)
o = torch.empty(M, 64, device=d, dtype=torch.bfloat16)
W = (0.05 * torch.randn(64, 576, device=d)).to(torch.bfloat16)
st = torch.zeros(64 * 128, device=d)
dW = torch.zeros(64, 576, device=d)
fl = 2.0 * M * 64 * 576


def timeit(fn, name, nbytes, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-36s %8.1f us  %7.1f TFLOP/s  %6.2f TB/s" % (name, ms * 1e3, fl / ms / 1e9, nbytes / ms / 1e9))


T = M * 64 * 2
timeit(lambda: lib.conv3x3_c64(x.data_ptr(), W.data_ptr(), o.data_ptr(), None, st.data_ptr(), N, H, H, 0, rt.stream()), "slab fwd + stats", 2 * T)
timeit(lambda: lib.conv3x3_c64(y.data_ptr(), W.data_ptr(), o.data_ptr(), x.data_ptr(), None, N, H, H, 1, rt.stream()), "slab bwd-data + res", 3 * T)
timeit(lambda: lib.conv3x3_c64(x.data_ptr(), W.data_ptr(), o.data_ptr(), None, None, N, H, H, 0, rt.stream()), "slab fwd plain", 2 * T)
timeit(lambda: lib.wgrad3x3_c64(x.data_ptr(), y.data_ptr(), dW.data_ptr(), N, H, H, rt.stream()), "slab wgrad", 2 * T)
