"""Stand-alone BatchNorm passes of the ResNet (avec_bn_apply_fwd, avec_bn_bwd_reduce, avec_bn_bwd_apply; bf16, channels-last) at the four stage sizes of the
B = 32 step: us per launch and GB/s over the bytes each pass has to move, with NSET rotating buffer sets (> the 256 MB last-level cache) inside one captured graph.
usage: PYTHONPATH=. python tools/bench_bn.py"""
import torch
import avec_amd
from avec_amd import runtime as rt
from avec_amd.lib import lib

STAGES = [(3200 * 22 * 22, 64), (3200 * 11 * 11, 128), (3200 * 6 * 6, 256), (3200 * 3 * 3, 512)]


def timed(fn, nset, reps=3):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(nset):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                for i in range(nset):
                    fn(i)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (3 * reps * nset)


def main():
    dev = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    print("%-22s %-34s %8s %8s" % ("M x C", "pass", "us", "GB/s"))
    for M, C in STAGES:
        mb = M * C * 2 / 1e6
        nset = max(2, int(900 / (4 * mb)) + 1)
        bufs = [[torch.randn(M, C, device=dev).bfloat16() for _ in range(5)] for _ in range(nset)]        # y, res / out, a, dout, dy
        masks = [torch.randint(0, 256, (M * C // 8,), device=dev, dtype=torch.uint8) for _ in range(nset)]
        ss = torch.randn(4 * C, device=dev)
        ss[3 * C:] = ss[3 * C:].abs() + 0.5
        gamma = torch.randn(C, device=dev)
        dstats = torch.zeros(2 * C, device=dev)
        st = rt.stream
        cases = [
            ("apply fwd (relu)", 2, lambda i: lib.bn_apply_fwd(rt.dt(), bufs[i][0].data_ptr(), ss.data_ptr(), None, 2, bufs[i][2].data_ptr(), M, C, st())),
            ("apply fwd (+res, relu)", 3, lambda i: lib.bn_apply_fwd(rt.dt(), bufs[i][0].data_ptr(), ss.data_ptr(), bufs[i][1].data_ptr(), 2, bufs[i][2].data_ptr(), M, C, st())),
            ("bwd reduce (relu recomputed)", 2, lambda i: lib.bn_bwd_reduce(rt.dt(), bufs[i][3].data_ptr(), bufs[i][0].data_ptr(), None, ss.data_ptr(), 2, dstats.data_ptr(), M, C, st())),
            ("bwd reduce (relu from out)", 3, lambda i: lib.bn_bwd_reduce(rt.dt(), bufs[i][3].data_ptr(), bufs[i][0].data_ptr(), bufs[i][2].data_ptr(), ss.data_ptr(), 2, dstats.data_ptr(), M, C, st())),
            ("bwd apply (relu recomputed)", 3, lambda i: lib.bn_bwd_apply(rt.dt(), bufs[i][3].data_ptr(), bufs[i][0].data_ptr(), None, ss.data_ptr(), gamma.data_ptr(), dstats.data_ptr(), None, float(M), 2,
                                                                         bufs[i][4].data_ptr(), None, None, None, M, C, st())),
            ("bwd apply (out mask, + dres)", 5, lambda i: lib.bn_bwd_apply(rt.dt(), bufs[i][3].data_ptr(), bufs[i][0].data_ptr(), bufs[i][2].data_ptr(), ss.data_ptr(), gamma.data_ptr(), dstats.data_ptr(), None, float(M), 2,
                                                                           bufs[i][4].data_ptr(), bufs[i][1].data_ptr(), None, None, M, C, st())),
            ("apply fwd (+res, relu, bit mask)", 3, lambda i: lib.bn_apply_fwd_mask(rt.dt(), bufs[i][0].data_ptr(), ss.data_ptr(), bufs[i][1].data_ptr(), None, bufs[i][2].data_ptr(), masks[i].data_ptr(), M, C, st())),
            ("bwd reduce (bit mask)", 2, lambda i: lib.bn_bwd_reduce_mask(rt.dt(), bufs[i][3].data_ptr(), bufs[i][0].data_ptr(), masks[i].data_ptr(), ss.data_ptr(), dstats.data_ptr(), M, C, st())),
            ("bwd apply (bit mask, + dres)", 4, lambda i: lib.bn_bwd_apply_mask(rt.dt(), bufs[i][3].data_ptr(), bufs[i][0].data_ptr(), masks[i].data_ptr(), ss.data_ptr(), gamma.data_ptr(), dstats.data_ptr(), None, float(M),
                                                                                bufs[i][4].data_ptr(), bufs[i][1].data_ptr(), None, None, M, C, st())),
        ]
        for name, passes, fn in cases:
            us = timed(fn, nset)
            print("%-22s %-34s %8.1f %8.0f" % ("%d x %d" % (M, C), name, us, passes * mb * 1e6 / us / 1e3))
        del bufs


if __name__ == "__main__":
    main()
