"""compare the per-exchange local vectors of run directory B against run directory A (tools/gpu/r6_equiv8.sh): the first exchange of every rank whose LOCAL vector moved"""
import sys, torch
a_dir, b_dir, world = sys.argv[1], sys.argv[2], int(sys.argv[3])
for r in range(world):
    A, B = torch.load("%s/rank%d.pt" % (a_dir, r)), torch.load("%s/rank%d.pt" % (b_dir, r))
    assert len(A) == len(B)
    hits = []
    for i, ((ka, la, oa), (kb, lb, ob)) in enumerate(zip(A, B)):
        assert ka == kb, (ka, kb)
        d = (la - lb).abs().max().item() / (la.abs().max().item() + 1e-30)
        do = (oa - ob).abs().max().item() / (oa.abs().max().item() + 1e-30)
        if d > 1e-4 or do > 1e-4:
            hits.append((i, ka, d, do))
    if not hits or hits[0][0] > 40:
        continue                      # (the ResNet backward sums move a little in every run: atomics order on heavily cancelling sums)
    print("rank %d: %d of %d exchanges moved; first: %s" % (r, len(hits), len(A), [(i, k, "%.2e" % d, "%.2e" % do) for i, k, d, do in hits[:2]]))
    i0, k0 = hits[0][0], hits[0][1]
    la, lb = A[i0][1], B[i0][1]
    n = la.numel(); C = (n - 1) // 2 if n % 2 else n // 2
    dd = (la - lb).abs(); nzs = (dd > 1e-5 * la.abs().max()).nonzero().flatten().tolist()
    print("    local vector of #%d %s: n=%d C=%d differing %d (sum part %d, sumsq part %d, tail %d)" % (i0, k0, n, C, len(nzs), sum(1 for j in nzs if j < C), sum(1 for j in nzs if C <= j < 2 * C), sum(1 for j in nzs if j >= 2 * C)))
    for j in nzs[:6] + nzs[-3:]:
        print("      [%d] good %.7g bad %.7g  diff %.7g" % (j, la[j].item(), lb[j].item(), (lb[j] - la[j]).item()))
    torch.save({"good": A[i0:i0 + 2], "bad": B[i0:i0 + 2], "rank": r}, "gpurun_out/r06/equiv8_bad_%s_rank%d.pt" % (b_dir.split("_")[-1], r))
    import os
    if os.path.exists("%s/taps%d.pt" % (a_dir, r)):
        TA, TB = torch.load("%s/taps%d.pt" % (a_dir, r)), torch.load("%s/taps%d.pt" % (b_dir, r))
        for (na, ta), (nb, tb) in zip(TA, TB):
            dd = (ta - tb).abs()
            nz = (dd > 1e-5 * ta.abs().max()).nonzero()
            print("    tap %-9s shape %s max|a| %.4g  max|a-b| %.4g  differing %d%s" % (na, tuple(ta.shape), ta.abs().max().item(), dd.max().item(), nz.shape[0],
                  "" if not nz.shape[0] else "  first %s last %s  a %.6g b %.6g" % (nz[0].tolist(), nz[-1].tolist(), ta[tuple(nz[0])].item(), tb[tuple(nz[0])].item())))
