"""GPU (-m gpu), round-2 parity cases: the arithmetic and the kernel variants the benchmark actually runs.

* bf16 (the benchmarked dtype) full-model GRADIENTS against the fp64 oracle, per tensor, also with the kernel variants that only the B=32 shape
  selects (64-byte-row NT ring, deep TN splits) forced through the library's environment switches in a subprocess;
* SpecAugment (inside the timed region) as a property test on the device output (nnet/preprocessing.py:115-130);
* BASELINE config 5's shape (15 s clips: 240 000 samples / 376 frames) through the full AV model against the oracle;
* bench.py's own launcher (python bench.py --gpus 2 spawns its ranks) on one GPU over gloo.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests import bf16_grad_probe as probe

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRUCT_ZERO = ("key_layer.bias", "pos_layer.bias", "conv_module.layers.3.bias", "layers.0.0.bias")

# bf16 storage + bf16 MFMA inputs (8 mantissa bits, eps = 3.9e-3) through 24 conformer blocks / 17 convolutions with training-mode BatchNorm.
# The tolerance is CALIBRATED, per tensor, by what bf16 arithmetic itself costs on this network: the same oracle graph evaluated under torch's own
# bf16 autocast (CPU) against the fp64 oracle.  Measured (B = 2, this seed): conformer tensors 1-4 % (max ~10-15 %) in both; the ResNet front-end is
# ill-conditioned -- BatchNorm backward on 3x3 / 6x6 feature maps subtracts a dominant common mode: even the fp32 path jumps from 2e-4 to 3e-3 there -- and
# bf16 lands at 19 % (last block) .. 40-50 % (stem) with the HIP path and 22 % .. 67 % with torch autocast.  A wrong kernel (dropped term, wrong scale,
# transposed tile) gives O(1) errors on the tensors it touches AND everything upstream of them.
BF16_VS_AUTOCAST = 1.5          # HIP bf16 error <= 1.5 x torch-autocast bf16 error + 0.02, per tensor
BF16_GRAD_MEDIAN_TOL = 0.03
BF16_CONFORMER_MAX = 0.15


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _oracle_grads(sd0, dtype, autocast=False):
    from oracle import avec_oracle as O
    video, vlen, audio, alen, labels, llen = probe.av_inputs(2)
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        out = O.av_forward(sd, video.to(dtype), vlen, audio.to(dtype), alen, train=True, stats_out={})
        ls = O.total_loss({k: [v[0].float(), v[1]] for k, v in out.items()}, labels, llen, O.AV_LOSS_WEIGHTS)
    ls["loss"].backward()
    return {k: v.grad.clone() for k, v in sd.items() if v.requires_grad and v.grad is not None}, {k: float(v) for k, v in ls.items()}


@pytest.fixture(scope="module")
def av_oracle(tmp_path_factory):
    """fp64 oracle gradients of the full AV model at B = 2 (seed-0 init, the inputs of tests/test_gpu_parity.py), saved for the subprocess runs, and the
    per-tensor error of the same graph under torch's bf16 autocast (the calibration of the bf16 tolerance)"""
    model, sd0 = probe.build_model()
    g64, ref_losses = _oracle_grads(sd0, torch.float64)
    g16, _ = _oracle_grads(sd0, torch.float32, autocast=True)
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    e_auto = {k: l2(g16[k], g64[k]) for k in g64}
    path = str(tmp_path_factory.mktemp("oracle") / "g64.pt")
    torch.save(g64, path)
    return model, g64, ref_losses, path, e_auto


def _check_bf16(errs, losses, finite, ref_losses, e_auto, tag):
    assert finite, tag
    for k, v in ref_losses.items():
        assert abs(losses[k] - v) < 5e-2 * abs(v), (tag, k, losses[k], v)
    checked = sorted((e, k) for k, e in errs.items() if not k.endswith(STRUCT_ZERO))
    assert len(checked) > 800
    assert checked[len(checked) // 2][0] < BF16_GRAD_MEDIAN_TOL, (tag, checked[len(checked) // 2])
    for e, k in checked:
        assert e < BF16_VS_AUTOCAST * e_auto[k] + 0.02, (tag, k, e, e_auto[k])
        if "front_end" not in k:
            assert e < BF16_CONFORMER_MAX, (tag, k, e)


def test_full_model_bf16_grads_match_fp64_oracle(av_oracle):
    """every non-structurally-zero gradient tensor of the benchmarked arithmetic (bf16) against the fp64 truth"""
    model, g64, ref_losses, _, e_auto = av_oracle
    errs, losses, finite = probe.grad_errors(model, g64, "bf16")
    _check_bf16(errs, losses, finite, ref_losses, e_auto, "default kernels")


@pytest.mark.parametrize("env", [{"AVEC_NT_RB": "64", "AVEC_TN_WGS": "4096"}, {"AVEC_NT_RB": "128", "AVEC_TN_WGS": "64", "AVEC_TN_KT": "64"}],
                         ids=["rb64+deep_split", "rb128+shallow_split"])
def test_full_model_bf16_grads_bench_kernel_variants(av_oracle, tmp_path, env):
    """the same check with the NT ring variant (64-byte rows, chosen by tile count at B = 32) and the TN split depth forced"""
    _, _, ref_losses, g64_path, e_auto = av_oracle
    out = str(tmp_path / "errs.json")
    e = dict(os.environ, **env)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-m", "tests.bf16_grad_probe", g64_path, out], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    _check_bf16(res["errs"], res["losses"], res["finite"], ref_losses, e_auto, str(env))


# ----------------------------------------------------------------------------------------------
# SpecAugment
# ----------------------------------------------------------------------------------------------
def _runs(mask_1d):
    """[(start, length)] of the True runs of a 1-D bool tensor"""
    idx = mask_1d.nonzero().flatten().tolist()
    runs = []
    for i in idx:
        if runs and runs[-1][0] + runs[-1][1] == i:
            runs[-1][1] += 1
        else:
            runs.append([i, 1])
    return runs


def _coverable(runs, n_bands, max_width):
    """can the runs be the union of at most n_bands intervals of width <= max_width?"""
    if max_width <= 0:
        return not runs
    return sum(-(-ln // max_width) for _, ln in runs) <= n_bands


def test_specaugment_properties():
    """SpecAugment(mF=2, F=27, mT=5, pS=0.05) (nnet/preprocessing.py:115-130, torchaudio mask_along_axis semantics): on an all-ones input the zeros
    are exactly 2 batch-shared frequency bands of width < 27 (whole rows, every frame) plus, per sample, at most 5 time bands of width < int(0.05 len)
    inside [0, len); nothing else changes; time bands differ between samples; the draw is a function of (seed, step)."""
    import avec_amd
    from avec_amd import ops, runtime as rt
    mF, Fp, mT, pS = 2, 27, 5, 0.05
    B, NM, F = 8, 80, 400
    lens = torch.tensor([400, 400, 363, 250, 120, 57, 19, 400], device=dev())
    seen_freq, seen_time = set(), 0
    for trial in range(6):
        avec_amd.manual_seed(777 + trial)
        mel = torch.ones(B, NM, F, device=dev())
        out = ops.spec_augment_(mel.clone(), lens, mF, Fp, mT, pS, 5)
        again = ops.spec_augment_(mel.clone(), lens, mF, Fp, mT, pS, 5)
        assert torch.equal(out, again)                                   # same (seed, step, site) -> same masks
        assert ((out == 0) | (out == 1)).all()                           # masking only writes zeros
        z = (out == 0).cpu()
        # frequency bands: rows zeroed over ALL frames (also beyond the valid length), identical for every sample
        row_masked = z.all(dim=2)
        assert (row_masked == row_masked[0:1]).all()
        fr = _runs(row_masked[0])
        assert _coverable(fr, mF, Fp - 1), fr
        seen_freq.add(tuple(map(tuple, fr)))
        for b in range(B):
            ln = int(lens[b])
            free_rows = (~row_masked[b]).nonzero().flatten()
            if free_rows.numel() == 0:
                continue
            sub = z[b, free_rows]                                        # rows without a frequency mask: zeros are time masks only
            assert (sub == sub[0:1]).all()                               # a time mask covers every mel bin of its frames
            tr = _runs(sub[0])
            Tp = int(pS * ln)
            assert all(s + l <= ln for s, l in tr), (b, tr, ln)          # inside the valid length
            assert _coverable(tr, mT, Tp - 1), (b, tr, Tp)               # width = int(U * Tp) <= Tp - 1
            seen_time += len(tr)
        # different samples draw different time masks
        t0 = z[0, (~row_masked[0]).nonzero().flatten()[0]]
        t1 = z[1, (~row_masked[1]).nonzero().flatten()[0]]
        t7 = z[7, (~row_masked[7]).nonzero().flatten()[0]]
        assert not (torch.equal(t0, t1) and torch.equal(t0, t7))
        # the step counter changes the draw
        rt.advance_rng(dev())
        nxt = ops.spec_augment_(mel.clone(), lens, mF, Fp, mT, pS, 5)
        assert not torch.equal(out, nxt)
    assert len(seen_freq) > 1 and seen_time > 20                         # masks do occur and vary with the seed
    # lengths too short for a time mask (int(0.05 * 19) = 0): sample 6 has no time mask at all (checked by _coverable with width 0 above)


def test_specaugment_module_is_train_only():
    import nnet
    sa = nnet.SpecAugment(mF=2, F=27, mT=5, pS=0.05)
    x = torch.ones(2, 80, 100, device=dev())
    lens = torch.tensor([100, 60], device=dev())
    sa.eval()
    assert torch.equal(sa(x.clone(), lens), x)
    sa.train()
    y = sa(x.clone(), lens)
    assert (y == 0).any() or True          # (a draw of two zero-width bands is legal)
    assert ((y == 0) | (y == 1)).all()


# ----------------------------------------------------------------------------------------------
# BASELINE config 5 shape: 15 s clips (240 000 samples -> 1501 mel frames, 376 video frames)
# ----------------------------------------------------------------------------------------------
def test_full_model_config5_shape_matches_oracle():
    """AV model at the long-utterance shape (Ta = 240 000, Tv = 376, ragged second clip, 36 labels): the kernels that only long inputs select
    (alpha-in-LDS CTC, 12-tile / K-V time-sharing MFMA attention, 250-patch audio stage) -- 7 losses against the oracle, fp32 mode 1e-3, bf16 5e-2;
    output lengths bit-exact."""
    import avec_amd
    from oracle import avec_oracle as O
    model, sd0 = probe.build_model()
    torch.manual_seed(5)
    B = 2
    video = torch.randn(B, 376, 88, 88, 1)
    audio = 0.1 * torch.randn(B, 240000)
    vlen, alen = torch.tensor([376, 251]), torch.tensor([240000, 160000])
    labels, llen = torch.randint(1, 256, (B, 36)), torch.tensor([36, 22])
    with torch.no_grad():
        ref = O.av_forward(sd0, video, vlen, audio, alen, train=True, stats_out={})
        ref_losses = {k: float(v) for k, v in O.total_loss(ref, labels, llen, O.AV_LOSS_WEIGHTS).items()}
    d = dev()
    try:
        for dtype, tol in (("f32", 1e-3), ("bf16", 5e-2)):
            model.load_state_dict(sd0)
            avec_amd.set_compute_dtype(dtype)
            model.arena.zero_grad()
            losses, _, _, _ = model.forward_model([video.to(d), vlen.to(d), audio.to(d), alen.to(d)], (labels.to(d), llen.to(d)), compute_metrics=False)
            for k, v in ref_losses.items():
                assert abs(float(losses[k]) - v) < tol * abs(v), (dtype, k, float(losses[k]), v)
            losses["loss"].backward()
            assert torch.isfinite(model.arena.grad).all(), dtype
            if dtype == "f32":
                out = model([video.to(d), vlen.to(d), audio.to(d), alen.to(d)])
                for k in ref:
                    assert out[k][1].tolist() == [int(x) for x in ref[k][1]], k
                    assert list(out[k][0].shape) == list(ref[k][0].shape), k
    finally:
        avec_amd.set_compute_dtype("f32")


# ----------------------------------------------------------------------------------------------
# bench.py launches its own ranks
# ----------------------------------------------------------------------------------------------
def test_bench_self_spawns_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` (no torchrun) must start two ranks itself and print ONE JSON line from rank 0 (here: both ranks on cuda:0, gloo)."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--backend", "gloo", "--share-gpu",
                        "--no-cpu-baseline", "--no-kernel-timing"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["value"] > 0 and out["scaling"] == "weak"


# ----------------------------------------------------------------------------------------------
# harness paths the advisor flagged as untested: stale shadows, checkpoint round trip, gradient accumulation, clipping
# ----------------------------------------------------------------------------------------------
def _ao_model(seed=0):
    import nnet
    torch.manual_seed(seed)
    model = nnet.AudioEfficientConformerInterCTC(vocab_size=256, att_type="patch", interctc_blocks=[])
    for x in model.modules():
        if isinstance(x, torch.nn.Dropout):
            x.p = 0.0
        if hasattr(x, "drop_rate"):
            x.drop_rate = 0.0
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    model = model.to(dev()).train()
    model.encoder.spec_augment.eval()
    return model


def _ao_batch():
    torch.manual_seed(3)
    audio, alen = 0.1 * torch.randn(2, 16000), torch.tensor([16000, 12000])
    labels, llen = torch.randint(1, 256, (2, 5)), torch.tensor([5, 3])
    return [audio.to(dev()), alen.to(dev())], (labels.to(dev()), llen.to(dev()))


def test_arena_shadows_follow_weight_edits():
    """Weights changed after Model.to('cuda') by a SUB-MODULE load_state_dict (how the reference configs transplant the LRW front-end) or by an in-place
    edit must reach the compute-dtype GEMM shadows (advisor finding: the arena only refreshed after Model.load_state_dict / Adam)."""
    import avec_amd
    model = _ao_model().eval()
    inputs, _ = _ao_batch()
    for dtype in ("f32", "bf16"):
        avec_amd.set_compute_dtype(dtype)
        tol = 1e-5 if dtype == "f32" else 2e-2
        with torch.no_grad():
            y0 = model(inputs)["outputs"][0].float().clone()
            head = model.encoder.head
            head.load_state_dict({k: 2.0 * v for k, v in head.state_dict().items()})           # sub-module load: logits double
            y1 = model(inputs)["outputs"][0].float().clone()
            assert (y1 - 2.0 * y0).abs().max() <= tol * y0.abs().max() * 2, dtype
            head.weight.mul_(0.5)
            head.bias.mul_(0.5)                                                               # in-place edit (version counters): back to y0
            y2 = model(inputs)["outputs"][0].float().clone()
            assert (y2 - y0).abs().max() <= tol * y0.abs().max(), dtype
    avec_amd.set_compute_dtype("f32")


def test_grad_accumulation_clipping_and_checkpoint_roundtrip(tmp_path):
    import avec_amd
    avec_amd.set_compute_dtype("f32")
    inputs, targets = _ao_batch()
    # (1) two micro-steps of the same batch with accumulated_steps=2 == one plain step (loss / 2 summed twice; dropout off)
    a, b = _ao_model(), _ao_model()
    a.train_step(inputs, targets, precision=torch.float32)
    _, _, acc = b.train_step(inputs, targets, precision=torch.float32, accumulated_steps=2, acc_step=0)
    assert acc == 1 and int(b.model_step) == 0 and b.arena.grad.abs().max() > 0          # no optimizer step yet, gradients kept
    _, _, acc = b.train_step(inputs, targets, precision=torch.float32, accumulated_steps=2, acc_step=acc)
    assert acc == 0 and int(b.model_step) == 1
    # Adam's first moment is linear in the accumulated gradient (the parameters move by ~lr * sign(g) in step 1: a rounding-level difference in a near-zero
    # gradient flips a whole update, so they are compared through the moment)
    ea, eb = a.optimizer._flat["exp_avg"].double(), b.optimizer._flat["exp_avg"].double()
    assert ea.norm() > 0 and ((ea - eb).norm() / ea.norm()).item() < 1e-3
    assert (a.arena.master - _ao_model().arena.master).abs().max().item() > 0
    # (2) global-norm clipping of the flat arena
    g = torch.randn_like(a.arena.grad)
    a.arena.grad.copy_(g)
    n = a.clip_gradients(0.5)
    assert abs(float(n) - float(g.norm())) < 1e-3 * float(g.norm())
    assert abs(float(a.arena.grad.norm()) - 0.5) < 1e-3
    a.arena.grad.copy_(1e-3 * g / g.norm())
    a.clip_gradients(0.5)
    assert torch.allclose(a.arena.grad, 1e-3 * g / g.norm())                              # below the threshold: untouched
    a.arena.zero_grad()
    # (3) checkpoint round trip: parameters, Adam moments and the step counter; the next step continues identically
    path = str(tmp_path / "ckpt.ckpt")
    b.save(path)
    c = _ao_model(seed=9)
    c.load(path)
    assert int(c.model_step) == 1 and torch.equal(c.arena.master, b.arena.master)
    assert torch.equal(c.optimizer._flat["exp_avg"], b.optimizer._flat["exp_avg"]) and torch.equal(c.optimizer._flat["exp_avg_sq"], b.optimizer._flat["exp_avg_sq"])
    b.train_step(inputs, targets, precision=torch.float32)
    c.train_step(inputs, targets, precision=torch.float32)
    assert int(c.model_step) == 2
    ec, eb2 = c.optimizer._flat["exp_avg"].double(), b.optimizer._flat["exp_avg"].double()
    assert ((ec - eb2).norm() / eb2.norm()).item() < 1e-3
    # without the optimizer state the schedule restarts (nnet/model.py:527-536)
    e = _ao_model(seed=9)
    e.load(path, load_optimizer=False)
    assert int(e.model_step) == 0 and e.optimizer._flat["exp_avg"].abs().max() == 0
    sd_b = torch.load(path, map_location="cpu", weights_only=False)["model_state_dict"]
    for k, v in e.state_dict().items():
        if v.is_floating_point():
            assert torch.equal(v.cpu(), sd_b[k].cpu()), k


# ----------------------------------------------------------------------------------------------
# grouped parameter-gradient launches
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", ["64", "128"])
def test_gemm_tn_grouped_matches_torch(tile):
    """avec_gemm_tn_grouped: a mixed bag of weight-gradient products (FFN / QKV slice with ldp > I / strided-row residual conv / bias sums) in one launch
    against fp64 torch; accumulates into O (O starts non-zero)."""
    import ctypes
    from avec_amd.lib import BF16, TnItem, lib
    from avec_amd import runtime as rt
    torch.manual_seed(0)
    d = dev()
    shapes = [(3200, 1024, 256, None, False), (3200, 256, 1024, None, True), (1600, 360, 360, 1080, True), (1600, 1440, 360, None, True),
              (333, 64, 72, None, False), (800, 360, 256, None, True, 2), (70, 256, 256, None, True),
              # the 180-channel audio stage: rows of 360 / 1080 bytes (8-byte aligned only), widths that are not a multiple of 8, a 2-byte-aligned column slice
              (6400, 720, 180, None, True), (6400, 180, 720, None, True), (1613, 540, 180, None, True), (250, 180, 180, 541, False), (250, 180, 180, 544, True), (97, 44, 45, None, True)]
    if tile == "128":                            # enough 128x128 tiles in the group (>= 96) for the launcher to choose the big tile
        shapes.append((3200, 1440, 360, None, True))
    items, refs, outs, keep = [], [], [], []
    for sh in shapes:
        M, I, J, ldp, bias = sh[:5]
        step = sh[5] if len(sh) > 5 else 0
        ldp_ = ldp or I
        # (16 readable bytes behind the last row: rows that are not whole 16-byte chunks are fetched up to the next chunk boundary)
        P = torch.randn(M * ldp_ + 8, device=d).to(torch.bfloat16)[:M * ldp_].view(M, ldp_)
        Mq = M * step if step else M
        Q = torch.randn(Mq * J + 8, device=d).to(torch.bfloat16)[:Mq * J].view(Mq, J)
        O = torch.randn(I, J, device=d)
        bsum = torch.randn(I, device=d) if bias else None
        Pv = P[:, ldp_ - I:] if ldp else P                        # a column slice (e.g. the V third of dQ|dK|dV)
        Qv = Q[::step] if step else Q
        ref = O.double() + Pv.double().t() @ Qv.double()
        refb = (bsum.double() + Pv.double().sum(0)) if bias else None
        it = TnItem()
        it.P, it.Q, it.O, it.p_colsum = Pv.data_ptr(), Q.data_ptr(), O.data_ptr(), (bsum.data_ptr() if bias else None)
        it.ldp, it.ldq, it.ldo, it.M, it.I, it.J = ldp_, J, J, M, I, J
        it.q_rows_out, it.q_rows_in, it.q_step = (M, Mq, step) if step else (1, 1, 0)
        assert lib.raw("avec_gemm_tn_grouped_ok")(BF16, ctypes.byref(it)) == 1
        items.append(it); refs.append((ref, refb)); outs.append((O, bsum)); keep += [P, Q]
    lib.gemm_tn_grouped(BF16, (TnItem * len(items))(*items), len(items), rt.stream())
    torch.cuda.synchronize()
    for (ref, refb), (O, bsum), sh in zip(refs, outs, shapes):
        assert ((O.double() - ref).abs().max() / ref.abs().max()).item() < 2e-3, sh
        if refb is not None:
            assert ((bsum.double() - refb).abs().max() / refb.abs().max()).item() < 2e-3, sh
    # a row stride smaller than the width is refused
    bad = TnItem()
    P = torch.randn(64, 180, device=d).to(torch.bfloat16)
    bad.P, bad.Q, bad.O = P.data_ptr(), P.data_ptr(), outs[0][0].data_ptr()
    bad.ldp, bad.ldq, bad.ldo, bad.M, bad.I, bad.J = 176, 180, 180, 64, 180, 180
    assert lib.raw("avec_gemm_tn_grouped_ok")(BF16, ctypes.byref(bad)) == 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_layernorm_rows_and_grouped_param_grads(dtype):
    """dx-only LayerNorm backward (one wave per row) + avec_layernorm_param_grads_grouped against torch autograd for the conformer widths"""
    import avec_amd
    from avec_amd.lib import LnItem, lib
    from avec_amd import runtime as rt
    avec_amd.set_compute_dtype(dtype)
    d = dev()
    torch.manual_seed(1)
    items, checks, keep = [], [], []
    for M, D, with_res in [(3200, 256, True), (1600, 360, False), (6400, 180, True), (37, 1024, True), (130, 64, False)]:
        x = torch.randn(M, D, device=d) * 2 + 0.3
        g, b = torch.randn(D, device=d), torch.randn(D, device=d)
        dy32 = torch.randn(M, D, device=d)
        dy = dy32.to(rt.act_dtype())
        dres = torch.randn(M, D, device=d) if with_res else None
        xr = x.double().requires_grad_(True); gr = g.double().requires_grad_(True); br = b.double().requires_grad_(True)
        y = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
        y.backward(dy.double())
        mean, var = x.double().mean(1), x.double().var(1, unbiased=False)
        mean_f, rstd_f = mean.float(), (var + 1e-6).rsqrt().float()
        dx = torch.empty(M, D, device=d)
        lib.layernorm_bwd(rt.dt(), dy.data_ptr(), 0, x.data_ptr(), mean_f.data_ptr(), rstd_f.data_ptr(), g.data_ptr(), dx.data_ptr(),
                          dres.data_ptr() if with_res else None, None, None, M, D, rt.stream())
        ref_dx = xr.grad + (dres.double() if with_res else 0)
        tol = 1e-4 if dtype == "f32" else 1e-4          # dy is the same rounded tensor on both sides
        assert ((dx.double() - ref_dx).abs().max() / ref_dx.abs().max()).item() < tol, (M, D)
        dg, db = torch.randn(D, device=d), torch.randn(D, device=d)
        it = LnItem()
        it.dy, it.x, it.mean, it.rstd, it.dgamma, it.dbeta = dy.data_ptr(), x.data_ptr(), mean_f.data_ptr(), rstd_f.data_ptr(), dg.data_ptr(), db.data_ptr()
        it.M, it.D, it.dy_f32 = M, D, 0
        items.append(it)
        checks.append((dg, db, dg.double() + gr.grad, db.double() + br.grad, (M, D)))
        keep += [x, dy, mean_f, rstd_f]
    lib.layernorm_param_grads_grouped(rt.dt(), (LnItem * len(items))(*items), len(items), rt.stream())
    torch.cuda.synchronize()
    for dg, db, rg, rb, tag in checks:
        assert ((dg.double() - rg).abs().max() / rg.abs().max()).item() < 1e-4, tag
        assert ((db.double() - rb).abs().max() / rb.abs().max()).item() < 1e-4, tag
    avec_amd.set_compute_dtype("f32")


# ---- 3x3 convolution fast paths: shifted-window kernel (stride 1) and parity-class order (stride-2 backward-data) --------------------------
@pytest.mark.parametrize("Nimg,H,Cin,Cout,stride", [(7, 11, 128, 128, 1), (5, 6, 256, 256, 1), (9, 3, 512, 512, 1), (3, 22, 64, 64, 1), (2, 31, 32, 96, 1),
                                                    (3, 22, 64, 128, 2), (5, 11, 128, 256, 2), (7, 6, 256, 512, 2), (4, 7, 64, 64, 2)])
def test_conv3x3_fast_paths_match_torch_conv2d(Nimg, H, Cin, Cout, stride):
    """nnet/blocks.py:29-91 (ResNet 3x3 convolutions, NHWC here): forward + BatchNorm statistics epilogue and backward-data of the implicit-GEMM entry point
    `avec_gemm_nt` against torch.nn.functional.conv2d in fp32 on the CPU, for the shapes that select conv3x3_shift_kernel (stride 1, W <= 31; ragged last
    tile, windows crossing image and tensor ends) and the parity-class order (stride 2; even and odd image sizes)."""
    import torch.nn.functional as F
    import avec_amd
    from avec_amd import ops
    from avec_amd.lib import ROWS_CONV_FWD, ROWS_CONV_BWD
    avec_amd.set_compute_dtype("bf16")
    try:
        d, adt = dev(), torch.bfloat16
        g = torch.Generator().manual_seed(Nimg * 1000 + H)
        x = torch.randn(Nimg, H, H, Cin, generator=g).to(adt)
        Wt = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(adt)
        OH = (H - 1) // stride + 1
        M = Nimg * OH * OH
        dy = torch.randn(M, Cout, generator=g).to(adt)
        xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        yr = F.conv2d(xr, Wt.float(), stride=stride, padding=1)
        yr.backward(dy.float().reshape(Nimg, OH, OH, Cout).permute(0, 3, 1, 2))
        ref = yr.detach().permute(0, 2, 3, 1).reshape(M, Cout)
        dref = xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin)
        xd, dyd = x.to(d), dy.to(d)
        W = Wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(d)         # [Cout][kh][kw][Cin]
        Wb = Wt.permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous().to(d)        # [Cin][kh][kw][Cout]
        y = torch.full((M, Cout), float("nan"), device=d, dtype=adt)
        st = torch.zeros(64 * 2 * Cout, device=d)
        ops.gemm_nt(xd, W, y, M, Cout, 9 * Cin, rows=ops.rows_conv(H, H, Cin, 3, 3, stride, 1, OH, OH), mode=ROWS_CONV_FWD, stats=st)
        dx = torch.full((Nimg * H * H, Cin), float("nan"), device=d, dtype=adt)
        ops.gemm_nt(dyd, Wb, dx, Nimg * H * H, Cin, 9 * Cout, rows=ops.rows_conv(H, H, Cout, 3, 3, stride, 1, OH, OH), mode=ROWS_CONV_BWD)
        torch.cuda.synchronize()
        rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
        assert rel(y, ref) < 4e-3, rel(y, ref)                      # bf16 output rounding: 2^-9 rms
        assert rel(dx, dref) < 4e-3, rel(dx, dref)
        # every element individually (a wrong tap on one border pixel does not move the norm)
        assert float((y.float().cpu() - ref).abs().max()) < 0.02 * float(ref.abs().max()) + 1e-3
        assert float((dx.float().cpu() - dref).abs().max()) < 0.02 * float(dref.abs().max()) + 1e-3
        s = st.view(64, 2, Cout).sum(0).cpu()
        assert rel(s[0], ref.sum(0)) < 2e-3 and rel(s[1], (ref * ref).sum(0)) < 2e-3
    finally:
        avec_amd.set_compute_dtype("f32")


def test_conv3x3_fast_paths_equal_generic_kernels(tmp_path):
    """The same launches with the fast paths switched off (AVEC_NO_CONV_SHIFT / AVEC_NO_PERM2 / AVEC_NO_LEAN_CONV, read once per process): identical bf16 results up to the
    summation order (fp32 accumulation, one rounding)."""
    code = r'''
import sys, torch
sys.path.insert(0, %r)
import avec_amd
from avec_amd import ops
from avec_amd.lib import ROWS_CONV_FWD, ROWS_CONV_BWD
avec_amd.set_compute_dtype("bf16")
d = torch.device("cuda:0"); out = {}
for (Nimg, H, Cin, Cout, stride) in [(33, 11, 128, 128, 1), (33, 11, 128, 256, 2), (20, 22, 64, 128, 2), (61, 6, 256, 256, 1), (70, 3, 512, 512, 1)]:
    g = torch.Generator().manual_seed(H + stride)
    x = torch.randn(Nimg, H, H, Cin, generator=g).bfloat16().to(d); OH = (H - 1) // stride + 1; M = Nimg * OH * OH
    W = (torch.randn(Cout, 9 * Cin, generator=g) / 30).bfloat16().to(d); Wb = (torch.randn(Cin, 9 * Cout, generator=g) / 30).bfloat16().to(d)
    dy = torch.randn(M, Cout, generator=g).bfloat16().to(d)
    y = torch.empty(M, Cout, device=d, dtype=torch.bfloat16); dx = torch.empty(Nimg * H * H, Cin, device=d, dtype=torch.bfloat16)
    ops.gemm_nt(x, W, y, M, Cout, 9 * Cin, rows=ops.rows_conv(H, H, Cin, 3, 3, stride, 1, OH, OH), mode=ROWS_CONV_FWD)
    ops.gemm_nt(dy, Wb, dx, Nimg * H * H, Cin, 9 * Cout, rows=ops.rows_conv(H, H, Cout, 3, 3, stride, 1, OH, OH), mode=ROWS_CONV_BWD)
    out["y%%d_%%d" %% (H, stride)] = y.float().cpu(); out["dx%%d_%%d" %% (H, stride)] = dx.float().cpu()
torch.save(out, sys.argv[1])
''' % ROOT
    script = tmp_path / "run.py"
    script.write_text(code)
    res = {}
    for name, env in (("fast", {}), ("fast256", {"AVEC_SHIFT_BM": "256"}), ("generic", {"AVEC_NO_CONV_SHIFT": "1", "AVEC_NO_PERM2": "1", "AVEC_NO_LEAN_CONV": "1"})):
        e = dict(os.environ); e.update(env)
        p = tmp_path / (name + ".pt")
        subprocess.run([sys.executable, str(script), str(p)], check=True, env=e, timeout=600)
        res[name] = torch.load(p)
    for fast in ("fast", "fast256"):                 # (the 256-row tile of the shifted-window kernel is what the B = 32 step runs; small products pick 128 rows)
        for k in res[fast]:
            a, b = res[fast][k], res["generic"][k]
            assert torch.isfinite(a).all()
            assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max()), (fast, k)         # at most one bf16 ulp of the largest element
            assert float((a != b).float().mean()) < 0.05, (fast, k)                                    # and only where the fp32 sums straddle a rounding boundary


# ---- fp8 (OCP e4m3) forward Linear products: BASELINE config 5's arithmetic ----------------------------------------------------------------
def _e4m3_ref(x, amax):
    """torch's own e4m3 cast (round to nearest even) of x * (1 / scale), scale = amax * (1 / 448) in fp32 like the kernel, clamped like the kernel"""
    s = torch.tensor(max(float(amax), 1e-20), dtype=torch.float32) * torch.tensor(1.0 / 448.0, dtype=torch.float32)
    inv = (torch.tensor(1.0, dtype=torch.float32) / s).to(x.device)
    return (x.float() * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn), float(s)


@pytest.mark.parametrize("M,N,K", [(3200, 1024, 256), (800, 360, 1440), (37, 96, 64), (6400, 1080, 360), (130, 520, 2080)])
def test_fp8_quantizer_and_gemm_match_torch_e4m3(M, N, K):
    """avec_fp8_quantize is bit-exact against torch's float8_e4m3fn cast; avec_gemm_nt_fp8 equals the fp32 product of the dequantized operands
    (fp32 accumulation order only) through the bias + Swish + pre-activation epilogue of the FFN's first Linear (nnet/modules.py:257-289)."""
    from avec_amd.lib import lib, Epilogue, BF16, ACT_SWISH
    from avec_amd import runtime as rt
    import ctypes
    d = dev()
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 1.7).bfloat16().to(d)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(d)
    bias = torch.randn(N, generator=g).to(d)
    amax = torch.zeros(2, device=d)
    Kp = (K + 15) // 16 * 16                      # rows padded to whole 16-byte chunks (K = 360 -> 368), pad written as zeros
    q = torch.full((M, Kp), 0x55, dtype=torch.uint8, device=d)
    lib.fp8_quantize(BF16, x.data_ptr(), K, q.data_ptr(), Kp, M, K, amax.data_ptr(), 1, rt.stream())
    wq = torch.full((N, Kp), 0x55, dtype=torch.uint8, device=d)
    lib.fp8_quantize(0, w.data_ptr(), K, wq.data_ptr(), Kp, N, K, amax.data_ptr() + 4, 1, rt.stream())
    torch.cuda.synchronize()
    assert float(amax[0]) == float(x.float().abs().max()) and float(amax[1]) == float(w.abs().max())
    xr, sx = _e4m3_ref(x, amax[0]); wr, sw = _e4m3_ref(w, amax[1])
    assert torch.equal(q[:, :K].contiguous().view(torch.float8_e4m3fn).float(), xr.float()), "activation quantizer differs from torch's e4m3 cast"
    assert torch.equal(wq[:, :K].contiguous().view(torch.float8_e4m3fn).float(), wr.float())
    assert int(q[:, K:].max()) == 0 and int(wq[:, K:].max()) == 0 if Kp > K else True
    out = torch.empty(M, N, device=d, dtype=torch.bfloat16); pre = torch.empty_like(out)
    ep = Epilogue(); ep.out, ep.ldo, ep.out_f32 = out.data_ptr(), N, 0
    ep.out_pre, ep.ldpre, ep.bias, ep.act, ep.alpha = pre.data_ptr(), N, bias.data_ptr(), ACT_SWISH, 1.0
    lib.gemm_nt_fp8(q.data_ptr(), Kp, wq.data_ptr(), Kp, M, N, Kp, amax.data_ptr(), amax.data_ptr() + 4, ctypes.byref(ep), rt.stream())
    torch.cuda.synchronize()
    z = (xr.float().double() @ wr.float().double().t()) * (sx * sw) + bias.double()
    ref = z * torch.sigmoid(z)
    assert float((pre.double() - z).abs().max()) < 2.0 ** -8 * float(z.abs().max()) + 1e-6         # bf16 output rounding
    assert float((out.double() - ref).abs().max()) < 2.0 ** -7 * float(ref.abs().max()) + 1e-6


def test_fp8_full_model_config5_shape_losses():
    """BASELINE config 5 (15 s clips + fp8 GEMMs): the AV model with e4m3 operands in every eligible forward Linear against the fp32 oracle -- the seven losses
    within 5e-2 relative (the bf16 bound of test_full_model_config5_shape_matches_oracle), gradients finite and close to the bf16 run's; and the e4m3 path must
    actually have run: every weight slot and most activation slots carry a maximum."""
    import avec_amd
    from avec_amd import fp8
    from oracle import avec_oracle as O
    model, sd0 = probe.build_model()
    torch.manual_seed(5)
    B = 2
    video = torch.randn(B, 376, 88, 88, 1)
    audio = 0.1 * torch.randn(B, 240000)
    vlen, alen = torch.tensor([376, 251]), torch.tensor([240000, 160000])
    labels, llen = torch.randint(1, 256, (B, 36)), torch.tensor([36, 22])
    with torch.no_grad():
        ref = O.av_forward(sd0, video, vlen, audio, alen, train=True, stats_out={})
        ref_losses = {k: float(v) for k, v in O.total_loss(ref, labels, llen, O.AV_LOSS_WEIGHTS).items()}
    d = dev()
    grads = {}
    try:
        for mode in ("bf16", "fp8"):
            model.load_state_dict(sd0)
            avec_amd.set_compute_dtype("bf16")
            fp8.enable(mode == "fp8")
            model.arena.zero_grad()
            losses, _, _, _ = model.forward_model([video.to(d), vlen.to(d), audio.to(d), alen.to(d)], (labels.to(d), llen.to(d)), compute_metrics=False)
            for k, v in ref_losses.items():
                assert abs(float(losses[k]) - v) < 5e-2 * abs(v), (mode, k, float(losses[k]), v)
            losses["loss"].backward()
            torch.cuda.synchronize()
            assert torch.isfinite(model.arena.grad).all(), mode
            grads[mode] = model.arena.grad.clone()
        st = model.arena._fp8
        assert st.n >= 100, st.n                                            # ~19 blocks x (FFN 4 + QKV + out + pos + 2 pointwise) minus the 180-channel stage
        am = st.amax.cpu()
        assert (am[:st.n] > 0).all(), "a weight was not quantized"
        assert float((am[st.n:] > 0).float().mean()) > 0.8, "most eligible products must have run on e4m3 operands"
        # e4m3 operands perturb the gradients (through the activations the backward pass re-reads) but do not change their scale or direction
        a, b = grads["fp8"].double(), grads["bf16"].double()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos > 0.98 and 0.9 < float(a.norm() / b.norm()) < 1.1, (cos, float(a.norm() / b.norm()))
    finally:
        fp8.enable(False)
        avec_amd.set_compute_dtype("f32")


@pytest.mark.parametrize("M,N,K,lda", [(6400, 720, 180, None), (1613, 540, 180, None), (6400, 180, 540, None), (250, 180, 180, 542), (129, 64, 12, None), (37, 200, 180, 182),
                                       (3200, 256, 360, None)])
def test_gemm_nt_unaligned_rows_and_k_tail(M, N, K, lda):
    """Linear products of the 180-channel audio stage through the LDS-DMA kernel: rows that are only 8- / 2-byte aligned, K = 8n + 4 with the tail chunk fixed
    up in LDS (the 4 elements of the NEXT row it drags in must not reach the result; the last row of A and of W is fetched without reading behind the matrix:
    both matrices sit at the very end of their allocation here).  Against fp64 torch with the bias + residual epilogue of nnet/modules.py:257-289."""
    import avec_amd
    from avec_amd import ops
    avec_amd.set_compute_dtype("bf16")
    try:
        d = dev()
        g = torch.Generator().manual_seed(M + N + K)
        lda_ = lda or K
        A = torch.randn(M, lda_, generator=g).bfloat16().to(d)
        Av = A[:, lda_ - K:] if lda else A                       # a column slice: odd element offsets when lda - K is odd
        W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(d)
        bias = torch.randn(N, generator=g).to(d)
        res = torch.randn(M, N, generator=g).to(d)
        out = torch.full((M, N), float("nan"), device=d)
        ops.gemm_nt(Av, W, out, M, N, K, rows=ops.rows_plain(lda_), bias=bias, res=res, alpha=0.5, out_f32=True)
        torch.cuda.synchronize()
        ref = res.double() + 0.5 * (Av.double() @ W.double().t() + bias.double())
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, err                                   # fp32 accumulation of exact bf16 products: only the summation order differs
    finally:
        avec_amd.set_compute_dtype("f32")


def test_standalone_activations_match_torch():
    """nnet/activations.py:39-69: nnet.Swish / ReLU / GLU(dim=-1) called on their own (outside the fused composites) -- values and input gradients against torch"""
    import nnet
    d = dev()
    torch.manual_seed(3)
    for name, ref in (("Swish", lambda t: t * torch.sigmoid(t)), ("ReLU", torch.relu), ("GLU", lambda t: torch.nn.functional.glu(t, dim=-1))):
        x = torch.randn(5, 37, 64, device=d, requires_grad=True)
        y = nnet.activations.act_dict[name]()(x)
        g = torch.randn_like(y)
        y.backward(g)
        xr = x.detach().clone().requires_grad_(True)
        yr = ref(xr)
        yr.backward(g)
        assert y.shape == yr.shape and torch.allclose(y, yr, rtol=1e-5, atol=1e-6), name
        assert torch.allclose(x.grad, xr.grad, rtol=1e-5, atol=1e-6), name
