"""PMC target: a few launches of the stage-2 / stage-3 3x3 convolutions (forward) so that rocprofv3 --pmc rows can be read per dispatch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import ops
from avec_amd.lib import ROWS_CONV_FWD
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda"); adt = torch.bfloat16
def conv(Nimg, H, Cin, Cout, n=4):
    x = torch.randn(Nimg, H, H, Cin, device=d).to(adt); M = Nimg * H * H
    W = torch.randn(Cout, 9 * Cin, device=d).to(adt); y = torch.empty(M, Cout, device=d, dtype=adt)
    rows = ops.rows_conv(H, H, Cin, 3, 3, 1, 1, H, H)
    for _ in range(n): ops.gemm_nt(x, W, y, M, Cout, 9 * Cin, rows=rows, mode=ROWS_CONV_FWD)
    torch.cuda.synchronize()
conv(3200, 11, 128, 128)
conv(3200, 6, 256, 256)
