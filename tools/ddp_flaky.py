"""Repeat the 2-rank (one GPU, gloo) equivalence run and report, per parameter group, where it differs from the single-process gradients."""
import os, subprocess, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
env = dict(os.environ, MASTER_ADDR="127.0.0.1", AVEC_PEER_SYNCBN=os.environ.get("AVEC_PEER_SYNCBN", "1"))
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_equiv.py"), "--out", "/tmp/single.pt"], check=True, env=env, timeout=600)
a = torch.load("/tmp/single.pt")
print("config: peer", env["AVEC_PEER_SYNCBN"], "early", os.environ.get("AVEC_EARLY_ALLREDUCE", "1"), "runs", n, flush=True)
for it in range(n):
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(29540 + it),
                    os.path.join(ROOT, "tools", "ddp_equiv.py"), "--out", "/tmp/ddp.pt", "--backend", "gloo", "--share-gpu"], env=env, timeout=900,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    for l in r.stdout.splitlines():
        if "PEER ERR FLAG" in l:
            print("run", it, l, flush=True)
    if r.returncode != 0:
        print("run", it, "FAILED rc", r.returncode, "\n".join(l for l in r.stdout.splitlines() if "Error" in l or "error" in l or "assert" in l.lower() or "peer" in l)[-1500:], flush=True)
        continue
    b = torch.load("/tmp/ddp.pt")
    grp = collections.defaultdict(lambda: [0.0, 0.0])
    worst = []
    for k, (o, m) in a["names"].items():
        ga, gb = a["grad"][o:o + m].double(), b["grad"][o:o + m].double()
        key = ".".join(k.split(".")[:4])
        grp[key][0] += float((ga - gb).pow(2).sum()); grp[key][1] += float(ga.pow(2).sum())
        e = float((ga - gb).norm() / (ga.norm() + 1e-30))
        worst.append((e, k))
    bad = {k: (v[0] / max(v[1], 1e-30)) ** 0.5 for k, v in grp.items()}
    bad = {k: round(v, 4) for k, v in bad.items() if v > 5e-3 and "front_end" not in k}
    worst.sort(reverse=True)
    so = b.get("site_orders")
    same = so is None or all(o == so[0] for o in so)
    if not bad and same:
        continue
    if not same:
        diff = [(i, x, y) for i, (x, y) in enumerate(zip(so[0], so[1])) if x != y][:6]
        print("run", it, "SITE ORDER DIFFERS between ranks:", diff, flush=True)
    print("run", it, "peer", b["peer"], "loss", float(a["loss"]), float(b["loss"]), "bad groups:", bad, "worst:", [(round(e, 3), k) for e, k in worst[:4] if "front_end" not in k][:4], flush=True)
