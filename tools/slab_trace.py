"""Phase stamps of one workgroup of the stage-1 slab kernel (library built by tools/build_trace_c3s.sh, AVEC_LIB_PATH): cycles per phase and image."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import runtime as rt
from avec_amd.lib import lib
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda")
N, H = 3200, 22
M = N * H * H
x = torch.randn(N, H, H, 64, device=d).to(torch.bfloat16)
o = torch.empty(M * 64 + 512, device=d, dtype=torch.bfloat16)
y = torch.randn(M, 64, device=d).to(torch.bfloat16)
BWD = len(sys.argv) > 1 and sys.argv[1] == "bwd"
W = (0.05 * torch.randn(64, 576, device=d)).to(torch.bfloat16)
st = torch.zeros(64 * 128 + 8 * 16, device=d)
for _ in range(3):
    if BWD:
        lib.conv3x3_c64(y.data_ptr(), W.data_ptr(), o.data_ptr(), x.data_ptr(), None, N, H, H, 1, rt.stream())
    else:
        lib.conv3x3_c64(x.data_ptr(), W.data_ptr(), o.data_ptr(), None, st.data_ptr(), N, H, H, 0, rt.stream())
torch.cuda.synchronize()
t = (o[M * 64:M * 64 + 256].view(torch.float32) if BWD else st[8192:]).view(16, 8).cpu().long()
print("backward-data + residual" if BWD else "forward + statistics")
print("image |  wait+write  taps0-4  taps5-8  fold | total (core cycles) | 100 MHz ticks since previous image")
for k in range(13):
    r = t[k]
    nxt = t[k + 1][0] if k + 1 < 13 else r[4]
    dd = lambda a, b: int((b - a) % (1 << 24))
    print("%5d | %8d %8d %8d %8d | %8d | %s" % (k, dd(r[0], r[1]), dd(r[1], r[2]), dd(r[2], r[3]), dd(r[3], r[4]), dd(r[0], nxt), dd(t[k - 1][6], r[6]) if k else "-"))
