"""Microbenchmark of the column-reduction kernels (BatchNorm backward statistics) at the ResNet-18 front-end shapes.
    [AVEC_NO_TREE=1] [AVEC_COL8_BLOCKS=<cap>] python tools/bench_colreduce.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avec_amd.lib import lib, BF16
from avec_amd import runtime as rt

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
st = rt.stream()          # also registers the reduction workspace
for (M, C) in [(3200 * 484, 64), (3200 * 121, 128), (3200 * 36, 256), (3200 * 9, 512), (3200, 360)]:
    x = torch.randn(M, C, device=dev).bfloat16(); d = torch.randn(M, C, device=dev).bfloat16(); o = torch.relu(x)
    ss = torch.randn(4 * C, device=dev); ds = torch.zeros(2 * C, device=dev)
    for _ in range(3):
        lib.bn_bwd_reduce(BF16, d.data_ptr(), x.data_ptr(), o.data_ptr(), ss.data_ptr(), 2, ds.data_ptr(), M, C, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.bn_bwd_reduce(BF16, d.data_ptr(), x.data_ptr(), o.data_ptr(), ss.data_ptr(), 2, ds.data_ptr(), M, C, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("bn_bwd_reduce M=%8d C=%4d  %8.1f us  %6.2f TB/s" % (M, C, us, 3 * M * C * 2 / us / 1e6))
