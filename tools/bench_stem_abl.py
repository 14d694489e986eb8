"""stem3p forward kernel alone, timed with HIP events (AVEC_S3P_ABL read once per process)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import avec_amd
from avec_amd.lib import lib
dev = torch.device("cuda:0")
B, T, H, W = 32, 100, 88, 88
vb = torch.randn(B, T, H, W, device=dev).to(torch.bfloat16)
w8 = torch.zeros(64, 36, 8, device=dev, dtype=torch.bfloat16); w8[:, :35, 1:] = torch.randn(64, 35, 7, device=dev).to(torch.bfloat16) * 0.05
gamma = torch.ones(64, device=dev); bias = torch.zeros(64, device=dev)
zp = torch.empty(B * T, 22, 22, 64, device=dev, dtype=torch.bfloat16); idx = torch.empty(B * T, 22, 22, 64, device=dev, dtype=torch.uint8)
stats = torch.zeros(128 * 64, device=dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
lib.set_reduce_workspace(ws.data_ptr(), ws.numel())
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.stem3p_fwd(vb.data_ptr(), w8.data_ptr(), bias.data_ptr(), gamma.data_ptr(), zp.data_ptr(), idx.data_ptr(), stats.data_ptr(), B, T, H, W, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.stem3p_fwd(vb.data_ptr(), w8.data_ptr(), bias.data_ptr(), gamma.data_ptr(), zp.data_ptr(), idx.data_ptr(), stats.data_ptr(), B, T, H, W, st)
e1.record(); torch.cuda.synchronize()
print("ABL=%s  stem3p_fwd %.1f us (incl. col_finalize)" % (os.environ.get("AVEC_S3P_ABL", "0"), e0.elapsed_time(e1) * 100))
if int(os.environ.get("AVEC_S3P_ABL", "0")) & 64:
    t = stats[4096:4104].cpu().tolist()
    names = ["slab wait", "conv tile", "barrier 1", "ring write + stats", "barrier 2", "pool", "tiles"]
    tot = sum(t[:6])
    print("phases of workgroup 0 / wave 0, shader cycles (share): " + ", ".join("%s %.0f (%.0f%%)" % (n, v, 100 * v / tot) for n, v in zip(names[:6], t[:6])) + ", tiles %d, cycles/tile %.0f" % (t[6], tot / max(t[6], 1)))
