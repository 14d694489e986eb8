"""Grouped weight-gradient launch (avec_gemm_tn_grouped, bf16) on work-lists shaped like the conformer stacks' flushes, REPS launches inside a captured
graph: us per launch, TFLOP/s and the instance chosen.  AVEC_TNG_KT / AVEC_TNG_WGS / AVEC_TNG_TILE select the variant.
usage: python tools/bench_tn_grouped.py [reps]"""
import sys
import torch
import avec_amd
from avec_amd import runtime as rt
from avec_amd.lib import BF16, TnItem, lib

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def block(M, D, ff=4):
    """(M, I, J) of one conformer block's Linear / pointwise weight gradients"""
    return [(M, D * ff, D), (M, D, D * ff), (M, 3 * D, D), (M, D, D), (M, D, D), (M, 2 * D, D), (M, D, D), (M, D * ff, D), (M, D, D * ff)]


LISTS = {
    "visual 256 x 3200 rows, 2 blocks": (block(3200, 256) * 2)[:16],
    "audio 360 x 1600 rows, 2 blocks": (block(1600, 360) * 2)[:16],
    "audio 180 x 6400 rows, 2 blocks": (block(6400, 180) * 2)[:16],
    "fusion 360 x 800 rows, 2 blocks": (block(800, 360) * 2)[:16],
    "one block 256": block(3200, 256),
}


def main():
    dev = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    print("%-36s %5s %9s %9s  %s" % ("work-list", "items", "us", "TFLOP/s", "kernel"))
    for name, shapes in LISTS.items():
        items, keep, fl = [], [], 0.0
        for M, I, J in shapes:
            P = torch.randn(M * I + 8, device=dev).to(torch.bfloat16)[:M * I].view(M, I)
            Q = torch.randn(M * J + 8, device=dev).to(torch.bfloat16)[:M * J].view(M, J)
            O = torch.zeros(I, J, device=dev)
            bs = torch.zeros(I, device=dev)
            it = TnItem()
            it.P, it.Q, it.O, it.p_colsum = P.data_ptr(), Q.data_ptr(), O.data_ptr(), (bs.data_ptr() if I % 4 == 0 else None)
            it.ldp, it.ldq, it.ldo, it.M, it.I, it.J = I, J, J, M, I, J
            it.q_rows_out, it.q_rows_in, it.q_step = 1, 1, 0
            items.append(it); keep += [P, Q, O, bs]; fl += 2.0 * M * I * J
        arr = (TnItem * len(items))(*items)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                lib.gemm_tn_grouped(BF16, arr, len(items), rt.stream())
            kn = lib.raw("avec_last_kernel")()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(REPS):
                    lib.gemm_tn_grouped(BF16, arr, len(items), rt.stream())
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gr.replay()
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / (5 * REPS)
        print("%-36s %5d %9.1f %9.1f  %s" % (name, len(items), us, fl / us / 1e6, kn if isinstance(kn, str) else kn.decode()))


if __name__ == "__main__":
    main()
