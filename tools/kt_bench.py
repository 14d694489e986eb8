import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, avec_amd
from avec_amd import ops
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda")
for (M, N, K) in ((6400, 720, 180), (6400, 180, 720), (6400, 540, 180), (6400, 180, 540), (6400, 360, 180), (1600, 180, 180)):
    A = torch.randn(M, K, device=d).bfloat16(); W = torch.randn(N, K, device=d).bfloat16(); out = torch.empty(M, N, device=d, dtype=torch.bfloat16)
    for _ in range(5): ops.gemm_nt(A, W, out, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.gemm_nt(A, W, out, M, N, K)
    e1.record(); torch.cuda.synchronize()
    print("NO_KTAIL=%s M %d N %d K %d : %.2f us" % (os.environ.get("AVEC_NO_KTAIL"), M, N, K, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
