"""stem3p fused weight-gradient kernel alone, timed with HIP events (AVEC_S3W_ABL read once per process)"""
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
dp = torch.randn(B * T, 22, 22, 64, device=dev).to(torch.bfloat16); idx = torch.randint(0, 9, (B * T, 22, 22, 64), device=dev, dtype=torch.uint8)
ss = torch.ones(256, device=dev); dstats = torch.ones(128, device=dev)
dw = torch.zeros(64 * 245, device=dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
lib.set_reduce_workspace(ws.data_ptr(), ws.numel())
st = torch.cuda.current_stream().cuda_stream
def run():
    lib.stem3p_wgrad(vb.data_ptr(), w8.data_ptr(), bias.data_ptr(), dp.data_ptr(), idx.data_ptr(), ss.data_ptr(), gamma.data_ptr(), dstats.data_ptr(), None, 6195200.0,
                     dw.data_ptr(), None, None, B, T, H, W, st)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
print("ABL=%s  stem3p_wgrad %.1f us (incl. col_finalize)" % (os.environ.get("AVEC_S3W_ABL", "0"), e0.elapsed_time(e1) * 100))
