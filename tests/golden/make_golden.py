"""Generate the committed golden vectors by running THE REFERENCE (/root/reference, imported via
ref_shims) in the build container.  Output: tests/golden/*.npz + *.json (data only: inputs,
weights at reduced dims, expected outputs).  Re-run:  python tests/golden/make_golden.py
"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch
import ref_shims

nnet = ref_shims.import_reference()
torch.set_num_threads(4)


def nodrop(m):
    for x in m.modules():
        if isinstance(x, torch.nn.Dropout):
            x.p = 0.0
    return m


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out[f"{k}/{kk}"] = vv.detach().cpu().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        else:
            out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, sum(a.nbytes for a in out.values()) // 1024, "KiB")


def grads_of(module):
    return {k: p.grad.clone() for k, p in module.named_parameters()}


ATT = lambda cls, **kw: {"class": cls, "params": dict(num_heads=4, attn_drop_rate=0.0, num_pos_embeddings=64,
                                                      weight_init="default", bias_init="default", **kw)}
CONV = {"class": "Conv1d", "params": {"padding": "same", "kernel_size": 15}}


def conformer_block_case(name, D, De, T, stride, patch, lens, seed):
    torch.manual_seed(seed)
    att = ATT("RelPosPatch1dMultiHeadAttention", patch_size=patch) if patch > 1 else ATT("RelPos1dMultiHeadAttention")
    blk = nodrop(nnet.ConformerBlock(dim_model=D, dim_expand=De, ff_ratio=4, att_params=att, drop_rate=0.1,
                                     conv_stride=stride, conv_params=CONV)).train()
    # make BN / LN affine non-trivial
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
    B = len(lens)
    x = torch.randn(B, T, D, requires_grad=True)
    lengths = torch.tensor(lens)
    mask = nnet.Mask()(x, lengths)
    y = blk(x, mask=mask)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    save(name, x=x, lengths=lengths, mask=mask, y=y, w=w, dx=x.grad, sd=sd0, grads=grads_of(blk),
         sd_after=blk.state_dict(), meta=np.array([D, De, T, stride, patch, 4]))


def interctc_stack_case():
    torch.manual_seed(7)
    net = nodrop(nnet.ConformerInterCTC(dim_model=[32, 48], num_blocks=[2, 1], interctc_blocks=[1, 2], vocab_size=16,
                                        loss_prefix="x_ctc", att_params=[ATT("RelPosPatch1dMultiHeadAttention", patch_size=3),
                                                                          ATT("RelPos1dMultiHeadAttention")],
                                        conv_params=CONV, ff_ratio=4, drop_rate=0.1, mask=nnet.Mask(), conv_stride=2)).train()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(3, 23, 32)
    lengths = torch.tensor([23, 17, 5])
    y, ylen, inter = net(x, lengths)
    flat = {}
    for k, (lg, ln) in inter.items():
        flat[k + ".logits"] = lg
        flat[k + ".len"] = ln
    save("interctc_stack", x=x, lengths=lengths, y=y, ylen=ylen, inter=flat, sd=sd0)


def resnet_block_case(name, cin, cout, stride, seed):
    torch.manual_seed(seed)
    blk = nnet.ResNetBlock(in_features=cin, out_features=cout, kernel_size=(3, 3), stride=(stride, stride),
                           act_fun="ReLU", joined_post_act=True).train()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.randn(5, cin, 11, 11, requires_grad=True)
    y = blk(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    save(name, x=x, y=y, w=w, dx=x.grad, sd=sd0, grads=grads_of(blk), sd_after=blk.state_dict(),
         meta=np.array([cin, cout, stride]))


def visual_stem_case():
    torch.manual_seed(11)
    stem = torch.nn.Sequential(
        nnet.ConvNeuralNetwork(dim_input=1, dim_layers=64, kernel_size=(5, 7, 7), strides=(1, 2, 2), norm="BatchNorm3d",
                               act_fun="ReLU", drop_rate=0.0, dim=3),
        nnet.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding="same")).train()
    sd0 = {k: v.clone() for k, v in stem.state_dict().items()}
    x = torch.randn(2, 1, 6, 24, 24)
    y = stem(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    save("visual_stem", x=x, y=y, w=w, sd=sd0, grads=grads_of(stem), sd_after=stem.state_dict())


def audio_stem_case():
    torch.manual_seed(12)
    cnn = nnet.ConvNeuralNetwork(dim_input=1, dim_layers=180, kernel_size=3, strides=2, norm="BatchNorm2d", act_fun="Swish",
                                 drop_rate=0.0, dim=2).train()
    sd0 = {k: v.clone() for k, v in cnn.state_dict().items()}
    x = torch.randn(2, 1, 80, 37)
    lengths = torch.tensor([37, 20])
    y, ylen = cnn(x, lengths)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    save("audio_stem", x=x, lengths=lengths, y=y, ylen=ylen, w=w, sd=sd0, grads=grads_of(cnn))


def mel_case():
    torch.manual_seed(13)
    ap = nnet.AudioPreprocessing(16000, 512, 25, 10, 80, False, -5.6501, 4.2280)
    x = 0.1 * torch.randn(3, 4000)
    lengths = torch.tensor([4000, 3999, 1600])
    y, ylen = ap(x, lengths)
    save("mel_frontend", x=x, lengths=lengths, y=y, ylen=ylen)


def ctc_case():
    torch.manual_seed(14)
    loss = nnet.CTCLoss(zero_infinity=True, assert_shorter=False)
    B, T, V = 5, 14, 9
    logits = torch.randn(B, T, V, requires_grad=True)
    logit_len = torch.tensor([14, 10, 14, 3, 7])
    y = torch.randint(1, V, (B, 6))
    y[2, 1] = y[2, 0]  # repeated label
    y_len = torch.tensor([6, 4, 5, 6, 0])  # sample 3 infeasible (6 labels, 3 frames); sample 4 empty target
    l = loss((y, y_len), (logits, logit_len))
    l.backward()
    per = torch.nn.functional.ctc_loss(torch.log_softmax(logits, -1).transpose(0, 1), y, logit_len, y_len, blank=0,
                                       reduction="none", zero_infinity=True)
    save("ctc_loss", logits=logits, logit_len=logit_len, y=y, y_len=y_len, loss=l, per_utt=per, dlogits=logits.grad)


def small_modules_case():
    torch.manual_seed(15)
    ic = nnet.InterCTCResModule(dim_model=24, vocab_size=10)
    fu = nnet.FusionModule(a_dim_model=12, v_dim_model=12, f_dim_model=12)
    x = torch.randn(2, 5, 24)
    y, lg = ic(x)
    a, v = torch.randn(2, 5, 12), torch.randn(2, 5, 12)
    f = fu(a, v)
    save("small_modules", x=x, ic_y=y, ic_logits=lg, a=a, v=v, fu_y=f, ic_sd=ic.state_dict(), fu_sd=fu.state_dict())


def int_cases():
    out = {}
    alen = torch.tensor([16000, 63840, 240000, 40000, 12000, 1, 159, 160, 161, 640, 1279, 1280])
    l0 = torch.div(alen, 160, rounding_mode="floor") + 1
    chain = [l0]
    for _ in range(3):
        chain.append(torch.div(chain[-1] - 1, 2, rounding_mode="floor") + 1)
    out["audio_len"] = alen.tolist()
    out["length_chain"] = [c.tolist() for c in chain]
    x = torch.zeros(3, 8, 4)
    m = nnet.Mask()(x, torch.tensor([8, 5, 1]))
    out["mask_T8"] = m.tolist()
    att = nnet.RelPosPatch1dMultiHeadAttention(dim_model=4, num_heads=1, patch_size=3, num_pos_embeddings=16, attn_drop_rate=0.0)
    Q, K, V, mp, padding = att.pad(x, x, x, m, chunk_size=3)
    mp = mp.squeeze(1)
    mp = -att.mask_pool(-mp).transpose(1, 2)
    mp = -att.mask_pool(-mp).transpose(1, 2).unsqueeze(1)
    out["patch_mask_T8_P3"] = mp.tolist()
    out["patch_padding"] = int(padding)
    sch = nnet.NoamDecayScheduler(warmup_steps=10000, dim_decay=360, val_factor=2)
    out["noam_lr"] = {str(s): float(sch.get_val_step(s)) for s in [1, 2, 100, 9999, 10000, 10001, 50000]}
    json.dump(out, open(os.path.join(HERE, "int_cases.json"), "w"))
    print("wrote int_cases.json")


def adam_case():
    torch.manual_seed(16)
    p = torch.nn.Parameter(torch.randn(7, 5))
    opt = nnet.Adam(params=[p], lr=nnet.NoamDecayScheduler(10000, 360, 2), betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    p0 = p.detach().clone()
    gs, ps = [], []
    for _ in range(3):
        g = torch.randn(7, 5)
        p.grad = g.clone()
        opt.step()
        gs.append(g)
        ps.append(p.detach().clone())
    save("adam_steps", p0=p0, g=torch.stack(gs), p=torch.stack(ps))


def full_model_case():
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC()
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False),
                  loss_weights={"v_ctc_2": 0.5 / 3, "v_ctc_5": 0.5 / 3, "a_ctc_7": 0.5 / 3, "a_ctc_10": 0.5 / 3,
                                "f_ctc_1": 0.5 / 3, "outputs": 0.5})
    nodrop(model).train()
    sd = model.state_dict()
    info = {"state_dict": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
            "param_names": [k for k, _ in model.named_parameters()],
            "n_params": sum(p.numel() for p in model.parameters()),
            "param_checksums": {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()
                                if v.is_floating_point()}}
    torch.manual_seed(1)
    B = 2
    video = torch.randn(B, 100, 88, 88, 1)
    audio = 0.1 * torch.randn(B, 63840)
    vlen, alen = torch.tensor([100, 63]), torch.tensor([63840, 40000])
    labels = torch.randint(1, 256, (B, 20))
    llen = torch.tensor([20, 13])
    outputs = model([video, vlen, audio, alen])
    losses, _, _, _ = model.forward_model([video, vlen, audio, alen], (labels, llen), compute_metrics=False)
    info["losses"] = {k: float(v) for k, v in losses.items()}
    info["output_lengths"] = {k: v[1].tolist() for k, v in outputs.items()}
    info["output_shapes"] = {k: list(v[0].shape) for k, v in outputs.items()}
    info["logits_head"] = {k: v[0][0, :2, :6].tolist() for k, v in outputs.items()}
    info["input_seed"] = 1
    info["labels"] = labels.tolist()
    json.dump(info, open(os.path.join(HERE, "av_full_seed0.json"), "w"))
    print("wrote av_full_seed0.json")
    # AO model, BASELINE config 1 (B=2, 1 s audio)
    torch.manual_seed(0)
    ao = nodrop(nnet.AudioEfficientConformerInterCTC(vocab_size=256, att_type="patch", interctc_blocks=[])).eval()
    torch.manual_seed(2)
    a = 0.1 * torch.randn(2, 16000)
    o = ao([a, torch.tensor([16000, 12000])])
    ao_info = {"n_params": sum(p.numel() for p in ao.parameters()), "shape": list(o["outputs"][0].shape),
               "lengths": o["outputs"][1].tolist(), "logits_head": o["outputs"][0][0, :2, :6].tolist()}
    json.dump(ao_info, open(os.path.join(HERE, "ao_cfg1_seed0.json"), "w"))
    print("wrote ao_cfg1_seed0.json")


def lrw_case():
    """BASELINE config 1: the LRW word classifier (VisualEfficientConformerCE), synthetic 29x88x88 clips, seed-0 init, train-mode BatchNorm, dropout off"""
    torch.manual_seed(0)
    model = nnet.VisualEfficientConformerCE(vocab_size=500)
    model.compile()
    nodrop(model).train()
    sd = model.state_dict()
    info = {"state_dict": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
            "n_params": sum(p.numel() for p in model.parameters()),
            "param_checksums": {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items() if v.is_floating_point()}}
    torch.manual_seed(3)
    video = torch.randn(2, 1, 29, 88, 88)
    labels = torch.randint(0, 500, (2,))
    logits = model(video)
    losses, metrics, _, _ = model.forward_model(video, labels)
    losses["loss"].backward()
    info["logits_shape"] = list(logits.shape)
    info["logits_head"] = logits[:, :8].tolist()
    info["loss"] = float(losses["loss"])
    info["labels"] = labels.tolist()
    info["input_seed"] = 3
    info["grad_norms"] = {k: float(p.grad.double().norm()) for k, p in model.named_parameters() if p.grad is not None and k in
                          ("encoder.head.weight", "encoder.back_end.conformer_blocks.11.feed_forward_module2.layers.4.weight",
                           "encoder.front_end.3.head.1.weight", "encoder.front_end.0.layers.0.0.weight")}
    json.dump(info, open(os.path.join(HERE, "lrw_ce_seed0.json"), "w"))
    print("wrote lrw_ce_seed0.json")


def vo_case():
    """visual-only VisualEfficientConformerInterCTC (default InterCTC after blocks 3, 6, 9), seed-0 init, train-mode BatchNorm, dropout off, ragged lengths"""
    torch.manual_seed(0)
    model = nnet.VisualEfficientConformerInterCTC()
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    nodrop(model).train()
    sd = model.state_dict()
    info = {"state_dict": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()], "n_params": sum(p.numel() for p in model.parameters())}
    torch.manual_seed(5)
    video = torch.randn(2, 40, 88, 88, 1)
    vlen = torch.tensor([40, 27])
    labels, llen = torch.randint(1, 256, (2, 6)), torch.tensor([6, 4])
    outputs = model([video, vlen])
    losses, _, _, _ = model.forward_model([video, vlen], (labels, llen), compute_metrics=False)
    info["losses"] = {k: float(v) for k, v in losses.items()}
    info["output_lengths"] = {k: v[1].tolist() for k, v in outputs.items()}
    info["output_shapes"] = {k: list(v[0].shape) for k, v in outputs.items()}
    info["logits_head"] = {k: v[0][0, :2, :6].tolist() for k, v in outputs.items()}
    info["labels"] = labels.tolist()
    info["input_seed"] = 5
    json.dump(info, open(os.path.join(HERE, "vo_interctc_seed0.json"), "w"))
    print("wrote vo_interctc_seed0.json")


def video_input_case():
    """the reference's own NormalizeVideo / align_video_to_audio / TimeMaskSecond (nnet/transforms.py:40-52,108-126,169-180) on a small clip.
    torchaudio is absent: TimeMaskSecond's call to torchaudio.functional.mask_along_axis is served by the restatement in oracle/video_input.py, so this pins the
    reference's loop count and its "mean of the clip as masked so far" fill value, not torchaudio's interval draw."""
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import video_input as VO
    rt = importlib.import_module("nnet.transforms")
    torch.manual_seed(21)
    u8 = torch.randint(0, 256, (60, 10, 12, 1), dtype=torch.uint8)
    x = u8.permute(3, 0, 1, 2).to(torch.float32) / 255
    xn = rt.NormalizeVideo(mean=(0.5,), std=(0.5,))(x)                          # (1,T,H,W)
    aligned, pads = {}, []
    for ta in (60 * 640 - 1, 60 * 640 + 5, 63 * 640 + 17, 66 * 640):
        v = rt.align_video_to_audio(xn.permute(1, 2, 3, 0), torch.zeros(ta))
        first = int((v.flatten(1).abs().sum(1) > 0).nonzero()[0])
        pads.append([ta, v.shape[0], first])
    aligned = rt.align_video_to_audio(xn.permute(1, 2, 3, 0), torch.zeros(63 * 640 + 17))
    calls = []

    def masker(spec, mask_param, mask_value, axis):
        assert axis == 2 and spec.dim() == 4
        out, se = VO.mask_along_time(spec, mask_param, mask_value)
        calls.append(list(se) + [float(mask_value)])
        return out
    rt.torchaudio.functional.mask_along_axis = masker
    torch.manual_seed(22)
    tm = rt.TimeMaskSecond(T_second=0.4, num_mask_second=1.0, fps=25.0, mean_frame=True)
    masked = tm(xn.permute(2, 3, 0, 1).clone()).permute(2, 3, 0, 1)             # (1,T,H,W)
    save("video_input_ref", u8=u8, normalized=xn, aligned=aligned, pads=np.array(pads), masked=masked, mask_calls=np.array(calls), mask_seed=np.array([22]))


def streaming_case():
    """context-limited masks (nnet/attentions.py:656-733) and a conformer stack run with them (nnet/networks.py:271-298: the dense mask is strided
    with the blocks; the patch attention min-pools it, nnet/attentions.py:354-362)."""
    torch.manual_seed(31)
    x = torch.randn(3, 14, 4)
    lens = torch.tensor([14, 9, 3])
    cases = {}
    for name, kw in {"l3_r0": dict(left_context=3, right_context=0), "l5_r2": dict(left_context=5, right_context=2), "r1": dict(right_context=1),
                     "l4": dict(left_context=4), "l2_r0_s4": dict(left_context=2, right_context=0, mask_start=4)}.items():
        cases[name] = nnet.Mask(**kw)(x, lens)
        cases[name + "_nolen"] = nnet.Mask(**kw)(x)
    torch.manual_seed(32)
    net = nodrop(nnet.ConformerInterCTC(dim_model=[32, 48], num_blocks=[2, 1], interctc_blocks=[1], vocab_size=16, loss_prefix="s_ctc",
                                        att_params=[ATT("RelPosPatch1dMultiHeadAttention", patch_size=3), ATT("RelPos1dMultiHeadAttention")],
                                        conv_params={"class": "Conv1d", "params": {"padding": "causal", "kernel_size": 15}}, ff_ratio=4, drop_rate=0.1,
                                        mask=nnet.Mask(left_context=8, right_context=2), conv_stride=2)).train()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    xs = torch.randn(3, 23, 32, requires_grad=True)
    lengths = torch.tensor([23, 17, 5])
    y, ylen, inter = net(xs, lengths)
    w = torch.randn_like(y)
    ((y * w).sum() + sum((lg * lg).sum() for lg, _ in inter.values())).backward()
    flat = {}
    for k, (lg, ln) in inter.items():
        flat[k + ".logits"] = lg
        flat[k + ".len"] = ln
    # fully causal variant: causal relative positions (nnet/attentions.py:234-256, nnet/embeddings.py:136-145) + causal mask + causal conv
    torch.manual_seed(33)
    cnet = nodrop(nnet.ConformerInterCTC(dim_model=[32, 48], num_blocks=[2, 1], interctc_blocks=[2], vocab_size=16, loss_prefix="c_ctc",
                                         att_params=[ATT("RelPosPatch1dMultiHeadAttention", patch_size=3), ATT("RelPos1dMultiHeadAttention", causal=True)],
                                         conv_params={"class": "Conv1d", "params": {"padding": "causal", "kernel_size": 15}}, ff_ratio=4, drop_rate=0.1,
                                         mask=nnet.Mask(left_context=6, right_context=0), conv_stride=2)).train()
    csd = {k: v.clone() for k, v in cnet.state_dict().items()}
    cx = torch.randn(2, 19, 32)
    clen = torch.tensor([19, 12])
    cy, cylen, cinter = cnet(cx, clen)
    save("streaming_causal", x=cx, lengths=clen, y=cy, ylen=cylen, logits=cinter["c_ctc_1"][0], sd=csd)
    save("streaming_stack", masks=cases, mask_lens=lens, x=xs, lengths=lengths, y=y, ylen=ylen, inter=flat, sd=sd0, w=w, dx=xs.grad,
         grads=grads_of(net))


def grouped_case():
    """SURVEY a11 + the key/value cache of 8f rank 4: GroupedRelPosMultiHeadSelfAttention (nnet/attentions.py:556-650; group sizes 3 and 1 as AudioEfficientConformerEncoder
    (att_type="grouped") uses them, nnet/networks.py:389-392) inside a ConformerBlock (forward, input and parameter gradients), and the attention layer alone decoding
    two chunks with `hidden` (outputs, attention weights, cache)."""
    for name, D, T, G, lens, seed in (("block_grouped3", 36, 20, 3, [20, 14, 5], 41), ("block_grouped1", 32, 17, 1, [17, 9], 42), ("block_grouped2", 32, 15, 2, [15, 8], 43)):
        torch.manual_seed(seed)
        att = {"class": "GroupedRelPosMultiHeadSelfAttention", "params": {"num_heads": 4, "group_size": G, "attn_drop_rate": 0.0, "max_pos_encoding": 64, "causal": False}}
        blk = nodrop(nnet.ConformerBlock(dim_model=D, dim_expand=D, ff_ratio=4, att_params=att, drop_rate=0.1, conv_stride=1, conv_params=CONV)).train()
        with torch.no_grad():
            for k, p in blk.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))          # also makes u, v non-zero
        sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
        x = torch.randn(len(lens), T, D, requires_grad=True)
        lengths = torch.tensor(lens)
        mask = nnet.Mask()(x, lengths)
        y = blk(x, mask=mask)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        save(name, x=x, lengths=lengths, mask=mask, y=y, w=w, dx=x.grad, sd=sd0, grads=grads_of(blk), meta=np.array([D, D, T, 1, G, 4]))
    # two-chunk decoding with the key/value cache (inference)
    for name, D, G, T1, T2, seed in (("attn_hidden_g1", 32, 1, 9, 6, 44), ("attn_hidden_g3", 36, 3, 9, 7, 45)):
        torch.manual_seed(seed)
        att = nnet.GroupedRelPosMultiHeadSelfAttention(dim_model=D, num_heads=4, attn_drop_rate=0.0, max_pos_encoding=64, group_size=G, causal=False).eval()
        with torch.no_grad():
            att.u.add_(0.3 * torch.randn(D)); att.v.add_(0.3 * torch.randn(D))
            for p in att.parameters():
                if p.dim() == 1:
                    p.add_(0.05 * torch.randn_like(p))
            x1, x2 = torch.randn(2, T1, D), torch.randn(2, T2, D)
            o1, w1, h1 = att.forwardQKV(x1, x1, x1, mask=None, return_att_w=True)
            o2, w2, h2 = att.forwardQKV(x2, x2, x2, mask=None, return_att_w=True, hidden=h1)
        save(name, x1=x1, x2=x2, o1=o1, w1=w1, o2=o2, w2=w2, h1K=h1["K"], h1V=h1["V"], h2K=h2["K"], h2V=h2["V"], sd=att.state_dict(), meta=np.array([D, G, T1, T2, 4]))
    # the audio encoder with att_type="grouped": state_dict keys / shapes and the seeded initialisation
    torch.manual_seed(0)
    enc = nnet.AudioEfficientConformerEncoder(att_type="grouped", interctc_blocks=[])
    sd = enc.state_dict()
    json.dump({"keys": list(sd.keys()), "shapes": [list(v.shape) for v in sd.values()],
               "probe": {k: float(sd[k].double().sum()) for k in list(sd.keys())[::37] if sd[k].is_floating_point()}, "n_params": sum(p.numel() for p in enc.parameters())},
              open(os.path.join(HERE, "ao_grouped_state.json"), "w"))
    print("wrote ao_grouped_state.json")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "grouped":
        grouped_case()
        sys.exit(0)

    if len(sys.argv) > 1 and sys.argv[1] == "video_input":
        video_input_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "streaming":
        streaming_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vo":
        vo_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lrw":
        lrw_case()
        sys.exit(0)
    conformer_block_case("block_relpos", 32, 32, 20, 1, 1, [20, 13, 7], 1)
    conformer_block_case("block_patch", 32, 32, 20, 1, 3, [20, 16, 4], 2)
    conformer_block_case("block_strided_patch", 32, 48, 21, 2, 3, [21, 10], 3)
    conformer_block_case("block_strided_relpos", 40, 64, 16, 2, 1, [16, 9], 4)
    interctc_stack_case()
    resnet_block_case("resnet_block_s1", 8, 8, 1, 5)
    resnet_block_case("resnet_block_s2", 8, 16, 2, 6)
    visual_stem_case()
    audio_stem_case()
    mel_case()
    ctc_case()
    small_modules_case()
    int_cases()
    adam_case()
    full_model_case()
    video_input_case()
    streaming_case()
    grouped_case()
