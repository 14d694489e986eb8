"""GPU (-m gpu), round 4: parity AT THE BENCHMARK CONFIGURATION and for the rows that were only property-checked.

* `test_bench_config_graphed_step_matches_oracle`: the step exactly as bench.py builds it -- B = 32, bf16, two streams, ONE hipGraph (shadow refresh + forward + six CTC
  losses + backward + Adam) -- replayed on the seed-0 weights, against `O.av_forward` on the same batch.  Dropout 0 and SpecAugment off (stochastic sites have no oracle
  counterpart; SpecAugment is pinned separately below).  Tolerance: 2e-2 relative on each of the seven losses (bf16 operands, fp32 accumulation; the fp32 mode of the
  same path holds 1e-3, tests/test_gpu_parity.py), BatchNorm running statistics 3e-2 of their max-norm.
* `test_specaugment_masks_match_oracle_draw_for_draw`: the device kernel's counter-based draws are exported (avec_debug_rng_uniform) and replayed through the oracle's
  restatement of torchaudio's mask_along_axis (nnet/preprocessing.py:115-130): masks BIT-EXACT.
* `test_attn_mfma_bf16_matches_oracle`: the bf16 MFMA attention kernels (the ones the bench runs) against `O.rel_pos_attention` directly, head widths 45 / 64 / 90:
  output 3e-2, gradients 6e-2 (max-norm relative; bf16 operands and bf16-rounded probabilities against an fp32 CPU evaluation)."""
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_mode():
    import avec_amd
    yield
    avec_amd.set_compute_dtype("f32")


def _nodrop(m):
    for x in m.modules():
        if isinstance(x, torch.nn.Dropout):
            x.p = 0.0
        if hasattr(x, "drop_rate"):
            x.drop_rate = 0.0
    return m


# ----------------------------------------------------------------------------------------------
# a2: SpecAugment, draw level
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,NM,F,mF,Fp,mT,pS,lens", [(4, 80, 400, 2, 27, 5, 0.05, [400, 311, 57, 400]), (3, 80, 1501, 2, 27, 5, 0.05, [1501, 900, 20]),
                                                    (2, 40, 64, 3, 15, 2, 0.2, None), (1, 80, 400, 0, 27, 5, 0.05, [333])])
def test_specaugment_masks_match_oracle_draw_for_draw(B, NM, F, mF, Fp, mT, pS, lens):
    import avec_amd
    from avec_amd import ops, runtime as rt
    from avec_amd.lib import lib
    from oracle import avec_oracle as O
    avec_amd.manual_seed(4242)
    g = torch.Generator().manual_seed(3)
    mel = (torch.rand(B, NM, F, generator=g) + 0.5)                     # strictly positive: a zero can only come from a mask
    lens_t = None if lens is None else torch.tensor(lens, dtype=torch.int64)
    sid = 77
    for step in range(2):                                              # two RNG steps: the masks must move with the device {seed, step} pair
        x = mel.clone().to(dev())
        ops.spec_augment_(x, None if lens_t is None else lens_t.to(dev()), mF, Fp, mT, pS, sid)
        # the kernel's draw ids (include/avec_hip.h: avec_debug_rng_uniform)
        ids = [2 * q + w for q in range(mF) for w in (0, 1)] + [1000 + 64 * b + 2 * q + w for b in range(B) for q in range(mT) for w in (0, 1)]
        ids_t = torch.tensor(ids, dtype=torch.int64, device=dev())
        u = torch.empty(len(ids), dtype=torch.float32, device=dev())
        lib.debug_rng_uniform(rt.rng_state(dev()).data_ptr(), sid, ids_t.data_ptr(), len(ids), u.data_ptr(), rt.stream())
        torch.cuda.synchronize()
        table = dict(zip(ids, u.cpu().tolist()))
        assert all(0.0 <= v < 1.0 for v in table.values())

        def uniform(kind, b, q, which):
            return table[2 * q + which] if kind == "f" else table[1000 + 64 * b + 2 * q + which]

        ref = O.spec_augment(mel, lens_t, mF, Fp, mT, pS, uniform)
        got = x.cpu()
        assert torch.equal(got == 0, ref == 0), "mask differs at rng step %d" % step
        assert torch.equal(got, ref)                                   # unmasked cells untouched, masked cells exactly 0
        if step == 0:
            first = got.clone()
        rt.advance_rng(dev())
    assert (mF == 0 and mT == 0) or not torch.equal(first, got), "the draws did not move with the RNG step"


# ----------------------------------------------------------------------------------------------
# a9: the bf16 MFMA attention kernels against the oracle, directly
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,H,T,B", [(180, 4, 67, 3), (256, 4, 100, 3), (360, 4, 50, 3), (256, 4, 200, 2)])
def test_attn_mfma_bf16_matches_oracle(D, H, T, B):
    import avec_amd
    import nnet
    from oracle import avec_oracle as O
    torch.manual_seed(5)
    layer = nnet.RelPos1dMultiHeadAttention(D, H, 10000, 0.0)
    with torch.no_grad():
        for p in layer.parameters():                                   # biases are zero-initialised: make every term of the product carry signal
            if p.dim() == 1:
                p.normal_(0.0, 0.1)
    sd = {"m." + k: v.detach().clone() for k, v in layer.state_dict().items()}
    x = torch.randn(B, T, D)
    lens = torch.tensor([T, max(T - 13, 1), max(T // 2, 1)][:B], dtype=torch.int64)
    dy = torch.randn(B, T, D)
    # oracle (fp32 CPU)
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.rel_pos_attention(sdg, "m", xr, O.key_padding_mask(T, lens), H)
    (ref * dy).sum().backward()
    # device, bf16: lengths fast path -> attn_mfma_fwd / attn_mfma_bwd + the batched MFMA products
    avec_amd.set_compute_dtype("bf16")
    layer = layer.to(dev())
    xd = x.to(dev()).requires_grad_(True)
    from avec_amd import ops
    ops.lib.load()
    out = layer.fused(xd, None, None, lens.to(dev()), 0.0, 0, False)
    (out * dy.to(dev())).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(out.detach().float().cpu(), ref.detach()) < 3e-2, rel_err(out.detach().float().cpu(), ref.detach())
    assert rel_err(xd.grad.float().cpu(), xr.grad) < 6e-2, ("dx", rel_err(xd.grad.float().cpu(), xr.grad))
    for k, p in layer.named_parameters():
        g_ref = sdg["m." + k].grad
        if g_ref is None or g_ref.abs().max() == 0:
            continue
        if k in ("key_layer.bias", "pos_layer.bias"):                  # analytically zero (a per-query constant added to every score cancels in the softmax)
            assert p.grad.float().abs().max() < 2e-2 * sdg["m.query_layer.bias"].grad.abs().max(), k
            continue
        e = rel_err(p.grad.float().cpu(), g_ref)
        assert e < 6e-2, (k, e)


# ----------------------------------------------------------------------------------------------
# the benchmark configuration itself: B = 32, bf16, two streams, one hipGraph
# ----------------------------------------------------------------------------------------------
def test_bench_config_graphed_step_matches_oracle():
    import avec_amd
    import nnet
    import bench
    from oracle import avec_oracle as O
    avec_amd.set_compute_dtype("bf16")
    avec_amd.manual_seed(1234)
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC(vocab_size=256, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2])
    _nodrop(model)
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev()).train()
    model.encoder.audio_encoder.spec_augment.eval()
    B = 32
    inputs, targets = bench.synthetic_batch(B, dev(), seed=0)             # the bench's own batch
    step = model.make_graphed_train_step(inputs, targets, precision=torch.bfloat16, warmup=1)      # (one eager optimisation step, then the capture)
    # back to the seed-0 state: the captured step starts with the shadow refresh, so the replay below runs on exactly these weights and statistics
    model.load_state_dict(sd0)
    # ... and with Adam's first moment at zero: after ONE replay exp_avg = (1 - beta1) (g + wd p0), i.e. the Adam state IS the gradient the timed graph computed --
    # the B = 32 backward variants (pair weight gradients over 3 200 images, two-stage ring, grouped launches, two-stream capture) checked on the graph itself
    model.optimizer._flat["exp_avg"].zero_()
    model.optimizer._flat["exp_avg_sq"].zero_()
    losses = step()
    torch.cuda.synchronize()
    got = {k: float(v) for k, v in losses.items()}
    sd_after = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if "running_" in k}
    grp = model.optimizer.param_groups[0]
    beta1, wd = grp["betas"][0], grp["weight_decay"]
    g_hip = {k: (model.optimizer.state[p]["exp_avg"].detach().float().cpu() / (1.0 - beta1) - wd * sd0[k].float()) for k, p in model.named_parameters()}
    cpu_in = [t.cpu() for t in inputs]
    stats = {}

    def oracle_pass(autocast):
        sd = {k: v.clone() for k, v in sd0.items()}
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        st = {}
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            o = O.av_forward(sd, cpu_in[0], cpu_in[1], cpu_in[2], cpu_in[3], train=True, stats_out=st)
            ls = O.total_loss({k: [v[0].float(), v[1]] for k, v in o.items()}, targets[0].cpu(), targets[1].cpu(), O.AV_LOSS_WEIGHTS)
        ls["loss"].backward()
        return {k: v.grad.detach().clone() for k, v in sd.items() if v.requires_grad and v.grad is not None}, {k: v.detach() for k, v in ls.items()}, st

    g_ref, ref, stats = oracle_pass(False)          # fp32 oracle on the FULL bench batch (its own distance to fp64 is 2e-4 .. 3e-3, tests/test_gpu_round2.py)
    g_auto, _, _ = oracle_pass(True)                # the same graph under torch's bf16 autocast: what bf16 arithmetic itself costs, per tensor
    assert set(ref) <= set(got) and len(ref) == 7
    for k in ref:
        a, b = got[k], float(ref[k])
        assert abs(a - b) < 2e-2 * abs(b), (k, a, b)
    worst = 0.0
    for k, v in stats.items():
        if "running_" not in k:
            continue
        worst = max(worst, rel_err(sd_after[k], v))
    assert worst < 3e-2, worst
    # gradients of the timed graph: per tensor, relative L2 error <= 1.5 x (error of torch's CPU bf16 autocast of the oracle graph) + 0.02; median < 0.03; every
    # non-front-end tensor < 0.15 -- the calibrated bf16 bound of tests/test_gpu_round2.py; structurally zero biases (a per-query constant cancels in the softmax; a
    # bias in front of training-mode BatchNorm cancels in the mean) must be small against a live bias gradient of the same module family
    from tests.test_gpu_round2 import BF16_VS_AUTOCAST, BF16_GRAD_MEDIAN_TOL, BF16_CONFORMER_MAX, STRUCT_ZERO
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    errs = sorted((l2(g_hip[k], g_ref[k]), k) for k in g_ref if not k.endswith(STRUCT_ZERO))
    assert len(errs) > 800, len(errs)
    assert errs[len(errs) // 2][0] < BF16_GRAD_MEDIAN_TOL, errs[len(errs) // 2]
    for e, k in errs:
        assert e < BF16_VS_AUTOCAST * l2(g_auto[k], g_ref[k]) + 0.02, (k, e, l2(g_auto[k], g_ref[k]))
        if "front_end" not in k:
            assert e < BF16_CONFORMER_MAX, (k, e)
    for k in g_ref:
        if k.endswith(("key_layer.bias", "pos_layer.bias")):
            live = g_ref[k.rsplit(".", 2)[0] + ".query_layer.bias"].abs().max()
            assert g_hip[k].abs().max() < 1e-1 * live, (k, float(g_hip[k].abs().max()), float(live))      # (pure bf16 rounding noise of dS summed over 3 200 rows: 3-5 % of a live bias gradient, run to run)


# ----------------------------------------------------------------------------------------------
# two consecutive LayerNorms per launch (avec_layernorm_fwd2 / _bwd2, ops.LN_PAIR)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("M,D", [(100, 256), (37, 360), (64, 180), (5, 512)])
def test_layernorm_pair_kernels_match_two_launches_and_torch(M, D, dtype):
    """y1 = LN1(x) (nnet/blocks.py:267,303), h2 = LN2(y1) (nnet/modules.py:278) from ONE launch == the two single launches bit for bit (same arithmetic in the same
    order), and torch.nn.functional.layer_norm in fp32 to 1e-5; likewise backward (dx2 = dres2 + LN2'(dy2), dx1 = LN1'(dx2), prepared gradient of dx1)."""
    import avec_amd
    from avec_amd import ops, runtime as rt
    from avec_amd.lib import lib
    avec_amd.set_compute_dtype(dtype)
    d = dev()
    adt = rt.act_dtype()
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g).to(d)
    w1, b1, w2, b2 = [(torch.randn(D, generator=g) * 0.3 + (1.0 if i % 2 == 0 else 0.0)).to(d) for i in range(4)]
    eps1, eps2 = 1e-6, 1e-5
    f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=d)
    y1, m1, r1, m2, r2 = f32(M, D), f32(M), f32(M), f32(M), f32(M)
    h2 = torch.empty(M, D, dtype=adt, device=d)
    lib.layernorm_fwd2(rt.dt(), x.data_ptr(), w1.data_ptr(), b1.data_ptr(), eps1, y1.data_ptr(), m1.data_ptr(), r1.data_ptr(),
                       w2.data_ptr(), b2.data_ptr(), eps2, h2.data_ptr(), m2.data_ptr(), r2.data_ptr(), M, D, rt.stream())
    ya, ma, ra = ops.layernorm_fwd(x, w1, b1, M, D, True, eps1)
    hb, mb, rb = ops.layernorm_fwd(ya, w2, b2, M, D, False, eps2)
    torch.cuda.synchronize()
    assert torch.equal(y1, ya) and torch.equal(m1, ma) and torch.equal(r1, ra)
    assert torch.equal(h2, hb) and torch.equal(m2, mb) and torch.equal(r2, rb)
    ref1 = torch.nn.functional.layer_norm(x.cpu(), (D,), w1.cpu(), b1.cpu(), eps1)
    ref2 = torch.nn.functional.layer_norm(ref1, (D,), w2.cpu(), b2.cpu(), eps2)
    assert rel_err(y1.cpu(), ref1) < 1e-5 and rel_err(h2.float().cpu(), ref2) < (1e-5 if dtype == "f32" else 6e-3)
    # backward
    dy2 = torch.randn(M, D, generator=g).to(d).to(adt)
    dres2 = torch.randn(M, D, generator=g).to(d)
    dx2, dx1, prep = f32(M, D), f32(M, D), torch.empty(M, D, dtype=adt, device=d)
    lib.layernorm_bwd2(rt.dt(), dy2.data_ptr(), y1.data_ptr(), m2.data_ptr(), r2.data_ptr(), w2.data_ptr(), dres2.data_ptr(), dx2.data_ptr(),
                       x.data_ptr(), m1.data_ptr(), r1.data_ptr(), w1.data_ptr(), dx1.data_ptr(), prep.data_ptr(), 0.5, 0.0, None, 0, M, D, rt.stream())
    dxa, dxb, prepb = f32(M, D), f32(M, D), torch.empty(M, D, dtype=adt, device=d)
    lib.layernorm_bwd(rt.dt(), dy2.data_ptr(), int(dtype == "f32"), y1.data_ptr(), m2.data_ptr(), r2.data_ptr(), w2.data_ptr(), dxa.data_ptr(), dres2.data_ptr(), None, None, M, D, rt.stream())
    lib.layernorm_bwd_prep(rt.dt(), dxa.data_ptr(), 1, x.data_ptr(), m1.data_ptr(), r1.data_ptr(), w1.data_ptr(), dxb.data_ptr(), None, prepb.data_ptr(), 0.5, 0.0, None, 0, M, D, rt.stream())
    torch.cuda.synchronize()
    assert torch.equal(dx2, dxa) and torch.equal(dx1, dxb) and torch.equal(prep, prepb)
    xr = x.cpu().double().requires_grad_(True)
    y1r = torch.nn.functional.layer_norm(xr, (D,), w1.cpu().double(), b1.cpu().double(), eps1)
    y1r.retain_grad()
    h2r = torch.nn.functional.layer_norm(y1r, (D,), w2.cpu().double(), b2.cpu().double(), eps2)
    (h2r * dy2.cpu().double()).sum().backward(retain_graph=True)
    g2 = y1r.grad + dres2.cpu().double()
    xr.grad = None
    y1r.backward(g2)
    assert rel_err(dx2.cpu().double(), g2) < 1e-4 and rel_err(dx1.cpu().double(), xr.grad) < 1e-4


def test_layernorm_pair_hand_over_in_a_block_stack_matches_separate_launches():
    """two ConformerBlocks back to back (the first one's closing LayerNorm feeds the second one's first pre-norm): losses and every parameter gradient with the
    paired launches (ops.LN_PAIR) equal the separate launches bit for bit"""
    import avec_amd
    import nnet
    from avec_amd import ops
    avec_amd.set_compute_dtype("f32")
    d = dev()
    res, calls = {}, {True: 0, False: 0}
    orig = ops.layernorm_bwd_pair
    for pair in (True, False):
        ops.LN_PAIR = pair

        def counted(*a, _pair=pair, **k):
            calls[_pair] += 1
            return orig(*a, **k)
        ops.layernorm_bwd_pair = counted
        try:
            torch.manual_seed(3)
            net = nnet.ConformerInterCTC(dim_model=64, num_blocks=3, interctc_blocks=[], vocab_size=16,
                                         att_params={"class": "RelPos1dMultiHeadAttention", "params": {"num_heads": 4, "attn_drop_rate": 0.0, "num_pos_embeddings": 200}},
                                         conv_params={"class": "Conv1d", "params": {"padding": "same", "kernel_size": 15}}, drop_rate=0.0).to(d).train()
            x = torch.randn(2, 20, 64, generator=torch.Generator().manual_seed(5)).to(d).requires_grad_(True)
            y, _, _ = net(x, torch.tensor([20, 13], device=d))
            (y * y).sum().backward()
            torch.cuda.synchronize()
            res[pair] = (y.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu().clone() for k, p in net.named_parameters() if p.grad is not None})
        finally:
            ops.LN_PAIR = True
            ops.layernorm_bwd_pair = orig
    assert calls == {True: 2, False: 0}, calls          # blocks 0 -> 1 and 1 -> 2 take the paired launches, forward (else there is nothing to hand over) and backward
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) > 20
    for k in res[True][2]:
        assert torch.allclose(res[True][2][k], res[False][2][k], rtol=1e-5, atol=1e-6), k
