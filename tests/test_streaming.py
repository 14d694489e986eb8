"""Streaming variants: context-limited attention masks (nnet/attentions.py:656-733), causal depthwise-conv padding
(nnet/layers.py:148-156) and a ConformerInterCTC stack run with both, against vectors produced by the reference (tests/golden/streaming_stack.npz).
CPU: oracle == reference (fp32 1e-5), product Mask == reference bit-exact.  GPU (-m gpu): the HIP stack == reference within 1e-3 (fp32 mode)."""
import pytest
import torch

from oracle import avec_oracle as O
from tests.helpers import load_npz, prefixed, rel_err

MASKS = {"l3_r0": dict(left_context=3, right_context=0), "l5_r2": dict(left_context=5, right_context=2), "r1": dict(right_context=1),
         "l4": dict(left_context=4), "l2_r0_s4": dict(left_context=2, right_context=0, mask_start=4)}
ATT = lambda cls, **kw: {"class": cls, "params": dict(num_heads=4, attn_drop_rate=0.0, num_pos_embeddings=64, weight_init="default", bias_init="default", **kw)}
STRUCT_ZERO = ("key_layer.bias", "pos_layer.bias", "conv_module.layers.3.bias")


@pytest.mark.parametrize("name", sorted(MASKS))
def test_context_masks_bit_exact(name):
    import nnet
    g = load_npz("streaming_stack")
    x, lens = torch.zeros(3, 14, 4), g["mask_lens"]
    kw = MASKS[name]
    assert torch.equal(nnet.Mask(**kw)(x, lens), g["masks"][name])
    assert torch.equal(nnet.Mask(**kw)(x), g["masks"][name + "_nolen"])
    assert torch.equal(O.context_mask(14, lens, kw.get("left_context"), kw.get("right_context"), kw.get("mask_start", 0)), g["masks"][name])
    assert torch.equal(O.context_mask(14, None, kw.get("left_context"), kw.get("right_context"), kw.get("mask_start", 0)), g["masks"][name + "_nolen"])


def test_oracle_streaming_stack_matches_reference():
    g = load_npz("streaming_stack")
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in prefixed(g["sd"]).items()}
    x = g["x"].clone().requires_grad_(True)
    y, ylen, inter = O.conformer_interctc(sd, "m", x, g["lengths"], [2, 1], [1], "s_ctc", [3, 1], True, {}, context=(8, 2, 0), causal_conv=True)
    assert rel_err(y, g["y"]) < 2e-5 and torch.equal(ylen, g["ylen"])
    for k, (lg, ln) in inter.items():
        assert rel_err(lg, g["inter"][k + ".logits"]) < 2e-5 and torch.equal(ln, g["inter"][k + ".len"])
    ((y * g["w"]).sum() + sum((lg * lg).sum() for lg, _ in inter.values())).backward()
    assert rel_err(x.grad, g["dx"]) < 1e-4
    for k, gr in g["grads"].items():
        if k.endswith(STRUCT_ZERO):
            continue
        assert rel_err(sd["m." + k].grad, gr) < 2e-4, k


@pytest.mark.gpu
def test_streaming_stack_on_gpu_matches_reference():
    import nnet
    g = load_npz("streaming_stack")
    net = nnet.ConformerInterCTC(dim_model=[32, 48], num_blocks=[2, 1], interctc_blocks=[1], vocab_size=16, loss_prefix="s_ctc",
                                 att_params=[ATT("RelPosPatch1dMultiHeadAttention", patch_size=3), ATT("RelPos1dMultiHeadAttention")],
                                 conv_params={"class": "Conv1d", "params": {"padding": "causal", "kernel_size": 15}}, ff_ratio=4, drop_rate=0.1,
                                 mask=nnet.Mask(left_context=8, right_context=2), conv_stride=2)
    import avec_amd
    avec_amd.set_compute_dtype("f32")
    avec_amd.manual_seed(1234)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "drop_rate"):
            m.drop_rate = 0.0
    net.load_state_dict(g["sd"])
    net = net.to("cuda").train()
    x = g["x"].to("cuda").requires_grad_(True)
    y, ylen, inter = net(x, g["lengths"].to("cuda"))
    assert rel_err(y.detach().cpu(), g["y"]) < 1e-3 and torch.equal(ylen.cpu(), g["ylen"])
    for k, (lg, ln) in inter.items():
        assert rel_err(lg.detach().cpu(), g["inter"][k + ".logits"]) < 1e-3 and torch.equal(ln.cpu(), g["inter"][k + ".len"])
    ((y * g["w"].to("cuda")).sum() + sum((lg * lg).sum() for lg, _ in inter.values())).backward()
    torch.cuda.synchronize()
    assert rel_err(x.grad.cpu(), g["dx"]) < 2e-3
    for k, p in net.named_parameters():
        if k.endswith(STRUCT_ZERO):
            continue
        assert rel_err(p.grad.detach().cpu(), g["grads"][k]) < 3e-3, (k, rel_err(p.grad.detach().cpu(), g["grads"][k]))


def _causal_net():
    import nnet
    return nnet.ConformerInterCTC(dim_model=[32, 48], num_blocks=[2, 1], interctc_blocks=[2], vocab_size=16, loss_prefix="c_ctc",
                                  att_params=[ATT("RelPosPatch1dMultiHeadAttention", patch_size=3), ATT("RelPos1dMultiHeadAttention", causal=True)],
                                  conv_params={"class": "Conv1d", "params": {"padding": "causal", "kernel_size": 15}}, ff_ratio=4, drop_rate=0.1,
                                  mask=nnet.Mask(left_context=6, right_context=0), conv_stride=2)


def test_oracle_causal_stack_matches_reference():
    """causal relative positions: under a causal mask the reference's T-row table / skew reads the same E[i-j] as the full-context layout"""
    g = load_npz("streaming_causal")
    y, ylen, inter = O.conformer_interctc(prefixed(g["sd"]), "m", g["x"], g["lengths"], [2, 1], [2], "c_ctc", [3, 1], True, {}, context=(6, 0, 0), causal_conv=True)
    assert rel_err(y, g["y"]) < 2e-5 and torch.equal(ylen, g["ylen"]) and rel_err(inter["c_ctc_1"][0], g["logits"]) < 2e-5


def test_causal_attention_requires_a_causal_mask():
    import nnet
    _causal_net()
    with pytest.raises(AssertionError):
        nnet.ConformerInterCTC(dim_model=32, num_blocks=1, interctc_blocks=[], vocab_size=16, att_params=ATT("RelPos1dMultiHeadAttention", causal=True),
                               conv_params={"class": "Conv1d", "params": {"padding": "same", "kernel_size": 15}}, mask=nnet.Mask(left_context=4, right_context=1))


@pytest.mark.gpu
def test_causal_stack_on_gpu_matches_reference():
    import avec_amd
    g = load_npz("streaming_causal")
    avec_amd.set_compute_dtype("f32")
    avec_amd.manual_seed(1234)
    net = _causal_net()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "drop_rate"):
            m.drop_rate = 0.0
    net.load_state_dict(g["sd"])
    net = net.to("cuda").train()
    y, ylen, inter = net(g["x"].to("cuda"), g["lengths"].to("cuda"))
    assert rel_err(y.detach().cpu(), g["y"]) < 1e-3 and torch.equal(ylen.cpu(), g["ylen"])
    assert rel_err(inter["c_ctc_1"][0].detach().cpu(), g["logits"]) < 1e-3
