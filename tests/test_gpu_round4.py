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
    losses = step()
    torch.cuda.synchronize()
    got = {k: float(v) for k, v in losses.items()}
    sd_after = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if "running_" in k}
    cpu_in = [t.cpu() for t in inputs]
    stats = {}
    with torch.no_grad():
        out = O.av_forward(sd0, cpu_in[0], cpu_in[1], cpu_in[2], cpu_in[3], train=True, stats_out=stats)
        ref = O.total_loss(out, targets[0].cpu(), targets[1].cpu(), O.AV_LOSS_WEIGHTS)
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
