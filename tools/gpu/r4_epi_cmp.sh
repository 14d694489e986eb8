mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_epi_cmp.log
: > $L
AVEC_LIB_PATH=$PWD/tools/_bin/libavec_base.so PYTHONPATH=$PWD python tools/gpu/r4_epi_cmp.py /tmp/a.pt >> $L 2>&1
PYTHONPATH=$PWD python tools/gpu/r4_epi_cmp.py /tmp/b.pt >> $L 2>&1
python - >> $L <<'PY'
import torch
a, b = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
def rel(u, v): return ((u - v).norm() / (v.norm() + 1e-30)).item()
for k in a:
    r = rel(a[k], b[k])
    if r > 0: print(k, "old vs new rel", r, "n differing", int(((a[k] - b[k]).abs() > 0).sum()), "of", a[k].numel())
for lib, d in (("old", a), ("new", b)):
    print(lib, "mask vs no-mask: y", rel(d["y_1"], d["y_0"]), "dx", rel(d["dx_1"], d["dx_0"]))
    worst = sorted(((rel(d[k], d[k.replace("g_1_", "g_0_")]), k) for k in d if k.startswith("g_1_")), reverse=True)[:4]
    print("   worst parameter gradients:", worst)
PY
cat $L
