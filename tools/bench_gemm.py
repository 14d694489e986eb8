"""Micro-benchmark of the GEMM family on the model's shapes (run on the GPU box): TFLOP/s per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd
from avec_amd import ops, runtime as rt
from avec_amd.lib import lib, ROWS_CONV_FWD, ROWS_CONV_BWD

avec_amd.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "bf16")
d = torch.device("cuda")
adt = rt.act_dtype()


def timeit(fn, flops, name, iters=10):
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
    print("%-44s %8.3f ms  %8.1f TFLOP/s" % (name, ms, flops / ms / 1e9))


def plain(M, N, K):
    A = torch.randn(M, K, device=d).to(adt); W = torch.randn(N, K, device=d).to(adt)
    out = torch.empty(M, N, device=d, dtype=adt)
    timeit(lambda: ops.gemm_nt(A, W, out, M, N, K), 2.0 * M * N * K, "nt plain %dx%dx%d" % (M, N, K))
    P = torch.randn(M, N, device=d).to(adt); O = torch.zeros(N, K, device=d)
    timeit(lambda: ops.gemm_tn(P, A, O, M, N, K), 2.0 * M * N * K, "tn plain %dx%dx%d" % (M, N, K))


def conv(Nimg, H, Cin, Cout, stride=1):
    x = torch.randn(Nimg, H, H, Cin, device=d).to(adt)
    OH = (H - 1) // stride + 1
    M = Nimg * OH * OH
    W = torch.randn(Cout, 9 * Cin, device=d).to(adt)
    Wb = torch.randn(Cin, 9 * Cout, device=d).to(adt)
    y = torch.empty(M, Cout, device=d, dtype=adt)
    st = torch.zeros(64 * 2 * Cout, device=d)
    rows = ops.rows_conv(H, H, Cin, 3, 3, stride, 1, OH, OH)
    fl = 2.0 * M * Cout * 9 * Cin
    timeit(lambda: ops.gemm_nt(x, W, y, M, Cout, 9 * Cin, rows=rows, mode=ROWS_CONV_FWD, stats=st), fl, "conv fwd  %dx%d^2 %d->%d s%d" % (Nimg, H, Cin, Cout, stride))
    timeit(lambda: ops.gemm_nt(x, W, y, M, Cout, 9 * Cin, rows=rows, mode=ROWS_CONV_FWD), fl, "conv fwd (no stats)")
    dx = torch.empty(Nimg * H * H, Cin, device=d, dtype=adt)
    rb = ops.rows_conv(H, H, Cout, 3, 3, stride, 1, OH, OH)
    timeit(lambda: ops.gemm_nt(y, Wb, dx, Nimg * H * H, Cin, 9 * Cout, rows=rb, mode=ROWS_CONV_BWD), 2.0 * Nimg * H * H * Cin * 9 * Cout, "conv bwd-data")
    if lib.raw("avec_conv3x3_c64_supported")(H, H, Cin, Cout, 3, 3, stride) and adt == torch.bfloat16:
        timeit(lambda: lib.conv3x3_c64(x.data_ptr(), W.data_ptr(), y.data_ptr(), None, st.data_ptr(), Nimg, H, H, 0, rt.stream()), fl, "conv fwd  slab kernel (stats)")
        timeit(lambda: lib.conv3x3_c64(y.data_ptr(), Wb.data_ptr(), dx.data_ptr(), x.data_ptr(), None, Nimg, H, H, 1, rt.stream()), fl, "conv bwd-data slab kernel (+res)")
    dW = torch.zeros(Cout, 9 * Cin, device=d)
    timeit(lambda: ops.gemm_tn(y, x, dW, M, Cout, 9 * Cin, q_rows=rows, q_mode=ROWS_CONV_FWD), fl, "conv wgrad")
    if lib.raw("avec_conv3x3_c64_supported")(H, H, Cin, Cout, 3, 3, stride) and adt == torch.bfloat16:
        timeit(lambda: lib.wgrad3x3_c64(x.data_ptr(), y.data_ptr(), dW.data_ptr(), Nimg, H, H, rt.stream()), fl, "conv wgrad slab kernel")
    if lib.raw("avec_wgrad3x3_c128_supported")(H, H, Cin, Cout, 3, 3, stride) and adt == torch.bfloat16:
        timeit(lambda: lib.wgrad3x3_c128(x.data_ptr(), y.data_ptr(), dW.data_ptr(), Nimg, Cin, H, H, rt.stream()), fl, "conv wgrad slab kernel (wide)")


plain(4096, 4096, 4096)
plain(8192, 1024, 1024)
plain(6400, 720, 180)
plain(3200, 1024, 256)
plain(1600, 360, 1440)
conv(3200, 22, 64, 64)
conv(3200, 11, 128, 128)
conv(3200, 6, 256, 256)
conv(3200, 3, 512, 512)
conv(3200, 22, 64, 128, 2)
conv(3200, 11, 128, 256, 2)
conv(3200, 6, 256, 512, 2)
