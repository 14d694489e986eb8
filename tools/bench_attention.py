"""Relative-position attention core (avec_relpos_attention_fwd / _bwd row pass, bf16 MFMA kernels) at the model's shapes, timed inside a captured graph.
usage: PYTHONPATH=. python tools/bench_attention.py"""
import torch
import avec_amd
from avec_amd import ops, runtime as rt

SHAPES = [(32, 4, 100, 64), (32, 4, 50, 90), (32, 4, 25, 90), (32, 4, 200, 45), (32, 4, 100, 64), (32, 4, 13, 90)]      # B, H, T, d


def timed(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (5 * reps)


def main():
    dv = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    adt = torch.bfloat16
    print("%-18s %8s %8s   (us per launch; MFMA flops fwd = 4*B*H*T*T*d*1.5)" % ("B,H,T,d", "fwd", "bwd"))
    for B, H, T, d in SHAPES:
        D = H * d
        qkv = (0.5 * torch.randn(B * T, 3 * D, device=dv)).to(adt)
        e = (0.5 * torch.randn(2 * T - 1, D, device=dv)).to(adt)
        do_ = torch.randn(B * T, D, device=dv).to(adt)
        lens = torch.full((B,), T, dtype=torch.int64, device=dv)
        o = torch.zeros(B * T, D, dtype=adt, device=dv)
        lse = torch.zeros(B * H, T, 2, dtype=torch.float32, device=dv)
        a = ops._attn_args(qkv, e, lens, 1, None, o, lse, B, H, T, d, D, 0)
        fwd = timed(lambda: ops.lib.relpos_attention_fwd(rt.dt(), ops._byref(a), rt.stream()))
        dqkv = torch.zeros(B * T, 3 * D, dtype=adt, device=dv)
        de = torch.zeros(2 * T - 1, D, dtype=torch.float32, device=dv)
        a.dout = do_.data_ptr()
        a.dq, a.lddq = dqkv.data_ptr(), 3 * D
        a.dk, a.dv, a.ldd = dqkv.data_ptr() + D * 2, dqkv.data_ptr() + 2 * D * 2, 3 * D
        a.de, a.ldde = de.data_ptr(), D
        Tld, Rld = (T + 7) // 8 * 8, (2 * T - 1 + 7) // 8 * 8
        scratch = torch.zeros(2, B * H, T, Tld, dtype=adt, device=dv)
        a.pbuf, a.dsbuf, a.ldt = scratch.data_ptr(), scratch.data_ptr() + scratch[0].numel() * 2, Tld
        dsrel = torch.zeros(H, B * T, Rld, dtype=adt, device=dv)
        a.dsrel, a.ldr = dsrel.data_ptr(), Rld
        bwd = timed(lambda: ops.lib.relpos_attention_bwd(rt.dt(), ops._byref(a), rt.stream()))
        print("%-18s %8.1f %8.1f" % ("%d,%d,%d,%d" % (B, H, T, d), fwd, bwd))


if __name__ == "__main__":
    main()
