"""GPU (-m gpu): two data-parallel ranks (gloo between two processes sharing the one MI355X of the test box; the collectives are the
same calls RCCL serves on a multi-GPU node) reproduce the single-process gradients, loss and BatchNorm running statistics."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_equals_single_process(tmp_path):
    single, ddp = str(tmp_path / "single.pt"), str(tmp_path / "ddp.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_equiv.py"), "--out", single], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29533", os.path.join(ROOT, "tools", "ddp_equiv.py"), "--out", ddp, "--backend", "gloo", "--share-gpu"],
                   check=True, env=env, timeout=900)
    a, b = torch.load(single), torch.load(ddp)
    # the overlapped exchange really ran: (fusion + audio-visual encoder + head) and (audio encoder) ranges, disjoint, inside the arena
    assert len(b["early"]) == 2 and not a["early"], b["early"]
    (l0, h0), (l1, h1) = sorted(b["early"])
    assert 0 <= l0 < h0 <= l1 < h1 <= b["numel"]
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-4 * abs(float(a["loss"]))
    # same math, different fp32 summation order (atomics, per-rank partial sums).  The BatchNorm-heavy visual front-end is ill-conditioned
    # in fp32 (the reference itself is several % from an fp64 evaluation there, see test_full_model_grads_match_oracle): looser bound.
    fe_num = fe_den = rest_num = rest_den = 0.0
    for k, (o, n) in a["names"].items():
        ga, gb = a["grad"][o:o + n].double(), b["grad"][o:o + n].double()
        if "front_end" in k:
            fe_num += float((ga - gb).pow(2).sum()); fe_den += float(ga.pow(2).sum())
        else:
            rest_num += float((ga - gb).pow(2).sum()); rest_den += float(ga.pow(2).sum())
    assert (rest_num / rest_den) ** 0.5 < 2e-3, (rest_num / rest_den) ** 0.5
    assert (fe_num / fe_den) ** 0.5 < 6e-2, (fe_num / fe_den) ** 0.5
    assert torch.allclose(a["running_mean"], b["running_mean"], atol=1e-5) and torch.allclose(a["running_var"], b["running_var"], rtol=1e-4, atol=1e-6)
