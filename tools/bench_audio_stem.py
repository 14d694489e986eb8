"""The audio stem's three big launches (Conv2d(1 -> C, 3x3, stride 2) + BatchNorm + Swish on the mel spectrogram, nnet/networks.py audio front-end) in a captured
graph: us per launch.  usage: python tools/bench_audio_stem.py [reps]"""
import sys
import torch
import avec_amd
from avec_amd import runtime as rt
from avec_amd.lib import lib

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REPS):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (5 * REPS)


def main():
    avec_amd.set_compute_dtype("bf16")
    d = torch.device("cuda:0")
    B, NM, F, C = 32, 80, 400, 180
    Fo, To = (NM - 1) // 2 + 1, (F - 1) // 2 + 1
    mel = torch.randn(B, NM, F, device=d)
    w, bias = torch.randn(C, 9, device=d) * 0.2, torch.randn(C, device=d) * 0.1
    y = torch.empty(B * To, C * Fo, dtype=torch.bfloat16, device=d)
    a = torch.empty_like(y)
    da = torch.randn(B * To, C * Fo, device=d).bfloat16()
    stats = torch.zeros(64 * 2 * C, device=d); ss = torch.cat([torch.ones(C), torch.zeros(C), torch.zeros(C), torch.ones(C)]).to(d)
    gamma = torch.ones(C, device=d); dstats = torch.zeros(2 * C, device=d)
    dw, db, dg, dbt = torch.zeros(C, 9, device=d), torch.zeros(C, device=d), torch.zeros(C, device=d), torch.zeros(C, device=d)
    st = lambda: rt.stream()
    fwd = lambda: lib.audio_stem_conv_fwd(rt.dt(), mel.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), stats.data_ptr(), B, NM, F, C, st())
    act = lambda: lib.audio_stem_act_fwd(rt.dt(), y.data_ptr(), ss.data_ptr(), a.data_ptr(), B, NM, F, C, st())
    base = (da.data_ptr(), y.data_ptr(), mel.data_ptr(), ss.data_ptr(), gamma.data_ptr(), dstats.data_ptr(), None, float(B * To * Fo))
    red = lambda: lib.audio_stem_bwd(rt.dt(), *base, 0, None, None, None, None, B, NM, F, C, st())
    par = lambda: lib.audio_stem_bwd(rt.dt(), *base, 1, dw.data_ptr(), db.data_ptr(), dg.data_ptr(), dbt.data_ptr(), B, NM, F, C, st())
    for name, fn in (("conv fwd + statistics", fwd), ("act fwd", act), ("bwd reduce", red), ("bwd params", par)):
        print("%-24s %8.1f us" % (name, timed(fn)))


if __name__ == "__main__":
    main()
