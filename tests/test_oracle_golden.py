"""CPU: the oracle (oracle/avec_oracle.py) replayed against golden vectors produced by the REFERENCE
(tests/golden/make_golden.py).  Tolerance: logits/outputs 1e-5 rel (fp32 CPU both sides), index
work bit-exact."""
import math

import pytest
import torch

from oracle import avec_oracle as O
from tests.helpers import load_json, load_npz, prefixed, rel_err

TOL = 2e-5
STRUCT_ZERO = ("key_layer.bias", "pos_layer.bias", "conv_module.layers.3.bias")


def _grad_sd(sd):
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
        out[k] = v
    return out


@pytest.mark.parametrize("name", ["block_relpos", "block_patch", "block_strided_patch", "block_strided_relpos"])
def test_conformer_block(name):
    g = load_npz(name)
    D, De, T, stride, patch, H = [int(v) for v in g["meta"]]
    sd = _grad_sd(prefixed(g["sd"]))
    x = g["x"].clone().requires_grad_(True)
    stats = {}
    y, s = O.conformer_block(sd, "m", x, g["mask"], H, patch, True, stats)
    assert s == stride
    assert rel_err(y, g["y"]) < TOL
    (y * g["w"]).sum().backward()
    assert rel_err(x.grad, g["dx"]) < 1e-4
    for k, gr in g["grads"].items():
        if k.endswith(STRUCT_ZERO):  # analytically zero (softmax shift invariance / bias before BN)
            assert sd["m." + k].grad.abs().max() < 1e-4 and gr.abs().max() < 1e-4
            continue
        assert rel_err(sd["m." + k].grad, gr) < 2e-4, k
    for k, v in stats.items():
        assert torch.allclose(v.float(), g["sd_after"][k[2:]].float(), atol=1e-5), k
    # the mask the oracle builds from lengths is the reference's Mask() bit for bit
    assert torch.equal(O.key_padding_mask(T, g["lengths"]), g["mask"])


def test_interctc_stack():
    g = load_npz("interctc_stack")
    sd = prefixed(g["sd"])
    y, ylen, inter = O.conformer_interctc(sd, "m", g["x"], g["lengths"], [2, 1], [1, 2], "x_ctc", [3, 1], True, {})
    assert rel_err(y, g["y"]) < TOL
    assert torch.equal(ylen, g["ylen"])
    keys = sorted(k[:-7] for k in g["inter"] if k.endswith(".logits"))
    assert sorted(inter) == keys
    for k in keys:
        assert rel_err(inter[k][0], g["inter"][k + ".logits"]) < TOL
        assert torch.equal(inter[k][1], g["inter"][k + ".len"])


@pytest.mark.parametrize("name", ["resnet_block_s1", "resnet_block_s2"])
def test_resnet_block(name):
    g = load_npz(name)
    stride = int(g["meta"][2])
    sd = _grad_sd(prefixed(g["sd"]))
    x = g["x"].clone().requires_grad_(True)
    stats = {}
    y = O.resnet_block(sd, "m", x, stride, True, stats)
    assert rel_err(y, g["y"]) < TOL
    (y * g["w"]).sum().backward()
    assert rel_err(x.grad, g["dx"]) < 1e-4
    for k, gr in g["grads"].items():
        assert rel_err(sd["m." + k].grad, gr) < 2e-4, k
    for k, v in stats.items():
        assert torch.allclose(v.float(), g["sd_after"][k[2:]].float(), atol=1e-5), k


def test_visual_stem():
    g = load_npz("visual_stem")
    # oracle.visual_frontend runs stem + ResNet; check the stem part through its building blocks
    sd = _grad_sd(prefixed(g["sd"]))
    import torch.nn.functional as F
    x = F.pad(g["x"], (3, 3, 3, 3, 2, 2))
    x = F.conv3d(x, sd["m.0.layers.0.0.weight"], sd["m.0.layers.0.0.bias"], stride=(1, 2, 2))
    x = F.relu(O.batch_norm(sd, "m.0.layers.0.1", x, True, {}))
    y = F.max_pool3d(F.pad(x, (1, 1, 1, 1, 0, 0)), (1, 3, 3), (1, 2, 2))
    assert rel_err(y, g["y"]) < TOL
    (y * g["w"]).sum().backward()
    for k, gr in g["grads"].items():
        if gr.abs().max() < 1e-4:
            continue
        assert rel_err(sd["m." + k].grad, gr) < 5e-4, k


def test_audio_stem():
    g = load_npz("audio_stem")
    sd = {"m.subsampling_module." + k: v for k, v in g["sd"].items()}
    sd["m.linear.weight"] = torch.eye(7200)[:4]
    sd["m.linear.bias"] = torch.zeros(4)
    # reference ConvNeuralNetwork output is (B,180,40,T'); compare through the reshape used by the encoder
    y, ylen = O.audio_stem(sd, "m", g["x"][:, 0], g["lengths"], True, {})
    ref = g["y"].reshape(2, 7200, -1).transpose(1, 2)[..., :4]
    assert rel_err(y, ref) < TOL
    assert torch.equal(ylen, g["ylen"])


def test_mel_frontend():
    g = load_npz("mel_frontend")
    y, ylen = O.mel_frontend(g["x"], g["lengths"])
    assert torch.equal(ylen, g["ylen"])
    # log-mel: compare in the log domain with an absolute tolerance (values span [-20, 5])
    assert (y - g["y"]).abs().max() < 2e-3
    assert rel_err(y.exp(), g["y"].exp()) < 1e-4


def test_ctc_loss():
    g = load_npz("ctc_loss")
    per = O.ctc_nll(g["logits"], g["logit_len"], g["y"], g["y_len"])
    assert torch.allclose(per, g["per_utt"], rtol=1e-5, atol=1e-5)
    assert per[3] == 0.0  # infeasible alignment -> zero_infinity
    assert abs(per.mean().item() - g["loss"].item()) < 1e-5
    out = {"outputs": [g["logits"], g["logit_len"]]}
    l = O.total_loss(out, g["y"], g["y_len"], {"outputs": 1.0})["loss"]
    assert abs(l.item() - g["loss"].item()) < 1e-5


def test_small_modules():
    g = load_npz("small_modules")
    sd = prefixed(g["ic_sd"])
    lg = O.linear(sd, "m.proj_1", g["x"])
    y = g["x"] + O.linear(sd, "m.proj_2", lg.softmax(-1))
    assert rel_err(lg, g["ic_logits"]) < TOL and rel_err(y, g["ic_y"]) < TOL
    sd = prefixed(g["fu_sd"])
    f = O.linear(sd, "m.layers.2", O.swish(O.linear(sd, "m.layers.0", torch.cat([g["a"], g["v"]], -1))))
    assert rel_err(f, g["fu_y"]) < TOL


def test_int_cases():
    j = load_json("int_cases")
    alen = torch.tensor(j["audio_len"])
    l = O.mel_lengths(alen)
    assert l.tolist() == j["length_chain"][0]
    for ref in j["length_chain"][1:]:
        l = O.strided_lengths(l)
        assert l.tolist() == ref
    m = O.key_padding_mask(8, torch.tensor([8, 5, 1]))
    assert m.tolist() == j["mask_T8"]
    assert O.patch_pool_mask(m, 3).tolist() == j["patch_mask_T8_P3"]
    assert j["patch_padding"] == 1
    for s, v in j["noam_lr"].items():
        assert math.isclose(O.noam_lr(int(s)), v, rel_tol=1e-12)
    # Tv = Ta // 640 + 1 meets the audio branch at Ta // 1280 + 1
    for ta in [16000, 63840, 240000]:
        tv = O.video_frames_for_audio(ta)
        a = O.strided_lengths(O.strided_lengths(O.strided_lengths(O.mel_lengths(torch.tensor(ta)))))
        assert int(O.strided_lengths(torch.tensor(tv))) == int(a)


def test_greedy_decode_known_answers():
    V = 5
    seq = [0, 1, 1, 0, 1, 2, 2, 2, 0, 0, 3, 4, 4]
    logits = torch.full((2, len(seq), V), -5.0)
    for t, s in enumerate(seq):
        logits[0, t, s] = 5.0
        logits[1, t, s] = 5.0
    ids = O.greedy_decode_ids(logits, torch.tensor([len(seq), 6]))
    assert ids == [[1, 1, 2, 3, 4], [1, 1, 2]]


def test_adam_steps():
    g = load_npz("adam_steps")
    p = g["p0"].clone()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for s in range(3):
        p, m, v = O.adam_update(p, g["g"][s], m, v, s + 1, O.noam_lr(s + 1))
        assert torch.allclose(p, g["p"][s], rtol=1e-6, atol=1e-9)


def test_lrw_classifier_oracle_matches_reference_golden():
    """oracle.lrw_forward + softmax_cross_entropy on the seed-0 LRW classifier against the reference's own logits / loss
    (tests/golden/lrw_ce_seed0.json): pins the oracle for BASELINE config 1."""
    import nnet
    g = load_json("lrw_ce_seed0")
    torch.manual_seed(0)
    model = nnet.VisualEfficientConformerCE(vocab_size=500)      # seeded init == reference (tests/test_nnet_api.py)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(g["input_seed"])
    video = torch.randn(2, 1, 29, 88, 88)
    labels = torch.randint(0, 500, (2,))
    assert labels.tolist() == g["labels"]
    with torch.no_grad():
        logits = O.lrw_forward(sd, video, train=True, stats_out={})
    assert list(logits.shape) == g["logits_shape"]
    assert rel_err(logits[:, :8], torch.tensor(g["logits_head"])) < 1e-4
    assert abs(float(O.softmax_cross_entropy(logits, labels)) - g["loss"]) < 1e-4 * g["loss"]


def test_vo_interctc_oracle_matches_reference_golden():
    """oracle.vo_forward + CTC losses on the seed-0 visual-only InterCTC model against the reference's own outputs (tests/golden/vo_interctc_seed0.json)"""
    import nnet
    g = load_json("vo_interctc_seed0")
    torch.manual_seed(0)
    model = nnet.VisualEfficientConformerInterCTC()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()] == g["state_dict"]
    torch.manual_seed(g["input_seed"])
    video = torch.randn(2, 40, 88, 88, 1)
    vlen = torch.tensor([40, 27])
    labels, llen = torch.randint(1, 256, (2, 6)), torch.tensor([6, 4])
    assert labels.tolist() == g["labels"]
    with torch.no_grad():
        out = O.vo_forward(sd, video, vlen, train=True, stats_out={})
    total = 0.0
    for w, k in zip(O.VO_LOSS_WEIGHTS, out):          # positional mapping of the weight list
        lg, ln = out[k]
        assert list(lg.shape) == g["output_shapes"][k] and [int(x) for x in ln] == g["output_lengths"][k]
        assert rel_err(lg[0, :2, :6], torch.tensor(g["logits_head"][k])) < 1e-4
        l = float(O.ctc_nll(lg, ln, labels, llen).mean())
        assert abs(l - g["losses"]["loss_" + k]) < 1e-4 * g["losses"]["loss_" + k]
        total += w * l
    assert abs(total - g["losses"]["loss"]) < 1e-4 * g["losses"]["loss"]


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/nnet"), reason="needs the reference tree (build container only)")
def test_oracle_fullsize_vs_reference_itself():
    """The pin of the oracle at FULL size (61.7 M parameters): tests/golden/check_oracle_fullsize.py imports the reference, runs both on the same
    inputs and compares 7 losses (1e-5), all 965 gradient tensors (fp64-calibrated, structurally-zero ones bounded) and the BatchNorm running statistics."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "check_oracle_fullsize.py")], cwd=root, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "ORACLE FULL-SIZE CHECK: PASS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name", ["block_grouped3", "block_grouped1", "block_grouped2"])
def test_grouped_attention_block(name):
    """ConformerBlock with GroupedRelPosMultiHeadSelfAttention (group sizes 3 / 1 / 2) -- oracle against the reference's output and gradients"""
    g = load_npz(name)
    D, De, T, stride, G, H = [int(v) for v in g["meta"]]
    sd = _grad_sd(prefixed(g["sd"]))
    x = g["x"].clone().requires_grad_(True)
    y, s = O.conformer_block(sd, "m", x, g["mask"], H, 1, True, {}, group=G)
    assert rel_err(y, g["y"]) < TOL
    (y * g["w"]).sum().backward()
    assert rel_err(x.grad, g["dx"]) < 1e-4
    for k, gr in g["grads"].items():
        if k.endswith("conv_module.layers.3.bias") or (k.endswith(("key_layer.bias", "pos_layer.bias")) and gr.abs().max() < 1e-4):
            # analytically zero: bias before BatchNorm; key / position bias = a per-query constant on every score -- the key bias only when no frame was zero-padded
            # AFTER the projection (T % G == 0)
            assert sd["m." + k].grad.abs().max() < 1e-4 and gr.abs().max() < 1e-4
            continue
        assert rel_err(sd["m." + k].grad, gr) < 2e-4, k


@pytest.mark.parametrize("name", ["attn_hidden_g1", "attn_hidden_g3"])
def test_grouped_attention_hidden_cache(name):
    """the key/value cache of the Transformer-XL / grouped attention (nnet/attentions.py:506-512, :588-600): two chunks, outputs + attention weights + caches"""
    g = load_npz(name)
    D, G, T1, T2, H = [int(v) for v in g["meta"]]
    sd = prefixed(g["sd"])
    o1, w1, h1 = O.grouped_rel_pos_attention(sd, "m", g["x1"], None, H, G, return_hidden=True)
    o2, w2, h2 = O.grouped_rel_pos_attention(sd, "m", g["x2"], None, H, G, hidden=h1, return_hidden=True)
    for got, ref in ((o1, g["o1"]), (w1, g["w1"]), (o2, g["o2"]), (w2, g["w2"]), (h1["K"], g["h1K"]), (h1["V"], g["h1V"]), (h2["K"], g["h2K"]), (h2["V"], g["h2V"])):
        assert got.shape == ref.shape and rel_err(got, ref) < TOL
