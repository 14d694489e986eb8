"""CPU oracle for the AV Efficient Conformer hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch-CPU *restatement* of the reference algorithm (burchim/AVEC,
`/root/reference`), written functionally over a `state_dict` that uses the reference's own key
names.  It is the checker for the HIP path: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  The product (`avec_amd/`) never does.

Pinning: every function below is checked against the reference itself, imported in the build
container by `tests/golden/ref_shims.py`; `tests/golden/make_golden.py` dumps the vectors under
`tests/golden/*.npz|json` and `tests/test_oracle_golden.py` replays them (no reference needed).
The mel front-end restates *torchaudio* (un-vendored, unpinned: reference requirements.txt:2) from
its documented defaults -> "parity unpinned" at that single boundary.

Each function cites the reference file:line it follows.
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# integer / index arithmetic (bit-exact rows a20/a21 of SURVEY section 8)
# --------------------------------------------------------------------------------------------


def mel_lengths(audio_len, hop=160):
    """nnet/preprocessing.py:77  len = len // hop + 1"""
    return torch.div(audio_len, hop, rounding_mode="floor") + 1


def strided_lengths(lengths, stride=2):
    """nnet/modules.py:127-128, nnet/networks.py:302  len = (len - 1) // s + 1"""
    return torch.div(lengths - 1, stride, rounding_mode="floor") + 1


def video_frames_for_audio(audio_len):
    """nnet/transforms.py:169-180  Tv = Ta // 640 + 1"""
    return audio_len // 640 + 1


def key_padding_mask(T, lengths):
    """nnet/attentions.py:682-733: (B,1,T,T) float mask, 1 = keep; only keys are masked."""
    ar = torch.arange(T, device=lengths.device)
    keep = (ar[None, :] < lengths[:, None]).to(torch.float32)  # (B, T)
    return keep[:, None, None, :].expand(-1, 1, T, T).contiguous()


def context_mask(T, lengths, left_context=None, right_context=None, mask_start=0):
    """nnet/attentions.py:694-731: band j - i <= right_context, i - j <= left_context (None = unlimited), the leading mask_start x mask_start block forced
    open, then intersected with the key padding; (B or 1, 1, T, T) float, 1 = keep."""
    m = torch.ones(T, T)
    if right_context is not None:
        m = m.tril(diagonal=right_context)
    if left_context is not None:
        m = torch.minimum(m, torch.ones(T, T).triu(diagonal=-left_context))
    m[:mask_start, :mask_start] = 1
    if lengths is None:
        return m[None, None]
    return torch.minimum(m[None, None], key_padding_mask(T, lengths)[:, :, :1])


def patch_pool_mask(mask, P):
    """nnet/attentions.py:140-171,357-362: zero-pad to a multiple of P then min-pool PxP."""
    T = mask.shape[-1]
    pad = (P - T % P) % P
    m = F.pad(mask, (0, pad, 0, pad), value=0.0)
    Tp = (T + pad) // P
    m = m.reshape(m.shape[0], 1, Tp, P, Tp, P)
    return m.amin(dim=(3, 5))


def greedy_decode_ids(logits, lengths, blank=0):
    """nnet/decoders.py:97-120: argmax -> cut to length -> unique_consecutive -> drop blank."""
    out = []
    am = logits.argmax(dim=-1)
    for b in range(logits.shape[0]):
        seq = am[b, : int(lengths[b])].tolist()
        ids, prev = [], None
        for s in seq:
            if s != prev and s != blank:
                ids.append(s)
            prev = s
        out.append(ids)
    return out


# --------------------------------------------------------------------------------------------
# mel filterbank front-end (a1)
# --------------------------------------------------------------------------------------------


def mel_filterbank(n_freqs=257, f_min=0.0, f_max=8000.0, n_mels=80, sample_rate=16000):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') (documented formula)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)  # (n_freqs, n_mels)


def mel_frontend(audio, lengths=None, n_fft=512, win=400, hop=160, n_mels=80):
    """nnet/preprocessing.py:57-85: Spectrogram(512,400,160) power 2 -> MelScale(80) -> log(x+1e-9).

    Explicit framing + rFFT (not torch.stft) so that it is an independent statement of the STFT:
    center reflect-pad n_fft//2, periodic Hann(win) zero-padded (centred) to n_fft."""
    x = audio if audio.dtype == torch.float64 else audio.float()
    B, L = x.shape
    xp = F.pad(x[:, None, :], (n_fft // 2, n_fft // 2), mode="reflect")[:, 0]
    n_frames = L // hop + 1
    frames = xp.unfold(1, n_fft, hop)[:, :n_frames]  # (B, F, n_fft)
    w = torch.zeros(n_fft, dtype=x.dtype)
    left = (n_fft - win) // 2
    w[left:left + win] = torch.hann_window(win, periodic=True, dtype=x.dtype)
    spec = torch.fft.rfft(frames * w, dim=-1)
    power = spec.real ** 2 + spec.imag ** 2  # (B, F, 257)
    mel = power @ mel_filterbank(n_fft // 2 + 1, 0.0, 8000.0, n_mels, 16000).to(x.dtype)  # (B, F, 80)
    out = (mel + 1e-9).log().transpose(1, 2).contiguous()  # (B, 80, F)
    if lengths is None:
        return out
    return out, mel_lengths(lengths, hop)


# --------------------------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------------------------


def mask_along_axis_bounds(size, mask_param, u_value, u_min):
    """torchaudio.functional.mask_along_axis (un-vendored, version unpinned: reference requirements.txt:2; restated from the library's published algorithm, the
    non-iid form used by transforms.FrequencyMasking / TimeMasking with p = 1.0):
        value = rand(1) * mask_param;  min_value = rand(1) * (size - value);  mask_start = min_value.long();  mask_end = min_value.long() + value.long()
    -> [start, end) of the masked run.  u_value, u_min: the two uniforms in draw order.  fp32 arithmetic like torch's default dtype.  "parity unpinned" at the torchaudio
    boundary: this pins the DRAW -> MASK map of the kernel, not torch's generator stream."""
    value = torch.tensor(u_value, dtype=torch.float32) * mask_param
    min_value = torch.tensor(u_min, dtype=torch.float32) * (size - value)
    start = int(min_value.long())
    return start, start + int(value.long())


def spec_augment(mel, lengths, mF, F_param, mT, pS, uniform):
    """nnet/preprocessing.py:115-130 (SpecAugment.forward, training): mF frequency masks shared by the batch (FrequencyMasking(freq_mask_param=F, iid_masks=False) on the
    whole (B, n_mels, T) tensor), then per sample b: T_b = int(pS * lengths[b]) and mT time masks TimeMasking(time_mask_param=T_b) on samples[b:b+1, :, :lengths[b]];
    masked cells are set to 0.  `uniform(kind, b, q, which)` supplies the draw that the torchaudio call would take from the global generator: kind "f" | "t", which 0 = value,
    1 = min_value (so the device kernel's counter-based draws can be replayed here).  Returns the masked copy."""
    out = mel.clone()
    B, n_mels, T = out.shape
    for q in range(mF):
        s0, s1 = mask_along_axis_bounds(n_mels, F_param, uniform("f", 0, q, 0), uniform("f", 0, q, 1))
        out[:, s0:s1, :] = 0.0
    for b in range(B):
        L = int(lengths[b]) if lengths is not None else T
        Tb = int(torch.tensor(pS, dtype=torch.float32) * torch.tensor(L))            # python float * int64 tensor -> fp32 tensor -> int() (preprocessing.py:126)
        for q in range(mT):
            s0, s1 = mask_along_axis_bounds(L, Tb, uniform("t", b, q, 0), uniform("t", b, q, 1))
            out[b, :, :L][:, s0:s1] = 0.0
    return out


def swish(x):
    """nnet/activations.py:39-45"""
    return x * torch.sigmoid(x)


def batch_norm(sd, prefix, x, train, stats_out=None, eps=1e-5, momentum=0.1):
    """nnet/normalizations.py:42-170 (channels-first input, stats over all but dim 1)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if train:
        dims = [d for d in range(x.dim()) if d != 1]
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            stats_out[prefix + ".running_mean"] = (1 - momentum) * rm + momentum * mean.detach()
            stats_out[prefix + ".running_var"] = (1 - momentum) * rv + momentum * var.detach() * n / max(n - 1, 1)
            stats_out[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = rm, rv
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps) * w.view(shape) + b.view(shape)


def layer_norm(sd, prefix, x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def linear(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def rel_pos_table(T, D):
    """nnet/embeddings.py:101-158: rows ordered p = T-1 ... -(T-1); PE[2k]=sin, PE[2k+1]=cos."""
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)  # (2T-1, 1)
    inv = 10000 ** (2 * torch.arange(0, D // 2, dtype=torch.float32).unsqueeze(0) / D)
    ang = pos / inv
    pe = torch.zeros(2 * T - 1, D)
    pe[:, 0::2] = ang.sin()
    pe[:, 1::2] = ang.cos()
    return pe


def rel_pos_attention(sd, prefix, x, mask, H):
    """nnet/attentions.py:280-323 with rel_to_abs (:234-278) restated as direct (i-j) indexing:
    scores[i,j] = (Q_i.K_j + Q_i.E_{i-j}) / sqrt(d);  E row index r = (T-1) - (i-j)."""
    B, T, D = x.shape
    d = D // H
    q = linear(sd, prefix + ".query_layer", x).view(B, T, H, d).transpose(1, 2)
    k = linear(sd, prefix + ".key_layer", x).view(B, T, H, d).transpose(1, 2)
    v = linear(sd, prefix + ".value_layer", x).view(B, T, H, d).transpose(1, 2)
    e = linear(sd, prefix + ".pos_layer", rel_pos_table(T, D).to(x.dtype)).view(2 * T - 1, H, d).transpose(0, 1)  # (H,2T-1,d)
    s_k = q @ k.transpose(2, 3)
    s_all = q @ e.transpose(1, 2).unsqueeze(0)  # (B,H,T,2T-1)
    i = torch.arange(T).unsqueeze(1)
    j = torch.arange(T).unsqueeze(0)
    idx = (T - 1) - (i - j)  # (T,T)
    s_e = s_all.gather(3, idx.expand(B, H, T, T))
    scores = (s_k + s_e) / d ** 0.5
    if mask is not None:
        scores = scores + mask.logical_not() * -1e9
    p = scores.softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, D)
    return linear(sd, prefix + ".output_layer", o)


def grouped_pos_table(T, Th, D, G):
    """nnet/embeddings.py:160-216, full context: the rows the slice [max_len - T + G//2 - Th : max_len - G%2 + T - G//2) selects, i.e. the frame offsets
    T + Th - 1 - G//2 ... -(T - G//2 - 1) (for even G the table holds position 0 twice and so does the slice)."""
    hi, lo = T - 1 - G // 2 + Th, -(T - G // 2 - 1)
    if G % 2:
        pos = torch.arange(hi, lo - 1, -1, dtype=torch.float32)
    else:
        pos = torch.cat([torch.arange(hi, -1, -1, dtype=torch.float32), torch.arange(0, lo - 1, -1, dtype=torch.float32)])
    inv = 10000 ** (2 * torch.arange(0, D // 2, dtype=torch.float32).unsqueeze(0) / D)
    ang = pos.unsqueeze(1) / inv
    pe = torch.zeros(pos.shape[0], D)
    pe[:, 0::2] = ang.sin()
    pe[:, 1::2] = ang.cos()
    return pe


def grouped_rel_pos_attention(sd, prefix, x, mask, H, G, hidden=None, return_hidden=False):
    """GroupedRelPosMultiHeadSelfAttention.forwardQKV (nnet/attentions.py:579-650; G = 1 is RelPosMultiHeadSelfAttention, :491-554), full context:
    projections; optional key/value cache (the cache keeps all frames, the attention drops its first Th % G); zero padding to multiples of G AFTER the
    projections (:140-171; a missing mask becomes an all-zero one when keys were padded); Qu = Q + u, Qv = Q + v; G frames -> one token by reshape;
    scores[i,j] = (Qu_i.K_j + Qv_i.E_{(Th/G + i) - j}) / sqrt(G*D/H) with rel_to_abs (:417-489) restated as direct indexing; mask[::G, ::G]."""
    B, T, D = x.shape
    dh = G * D // H
    q, k, v = linear(sd, prefix + ".query_layer", x), linear(sd, prefix + ".key_layer", x), linear(sd, prefix + ".value_layer", x)
    new_hidden = {"K": k.detach(), "V": v.detach()}
    if hidden:
        new_hidden = {"K": torch.cat([hidden["K"], k], 1).detach(), "V": torch.cat([hidden["V"], v], 1).detach()}
        cut = hidden["K"].shape[1] % G
        k, v = torch.cat([hidden["K"][:, cut:], k], 1), torch.cat([hidden["V"][:, cut:], v], 1)
    pq, pk = (-T) % G, (-k.shape[1]) % G
    Tkv = k.shape[1]
    q, k, v = F.pad(q, (0, 0, 0, pq)), F.pad(k, (0, 0, 0, pk)), F.pad(v, (0, 0, 0, pk))
    if mask is not None:
        mask = F.pad(mask, (0, pk) if mask.shape[2] == 1 else (0, pq, 0, pk), value=0)      # (sic) the reference pads (last dim by padding_Q, rows by padding_KV); equal when Th = 0
    elif pk:
        mask = F.pad(q.new_zeros(B, 1, 1, Tkv), (0, pk), value=0)
    Tp, Tkp = T + pq, Tkv + pk
    Tg, Tkg = Tp // G, Tkp // G
    qu = (q + sd[prefix + ".u"]).reshape(B, Tg, H, dh).transpose(1, 2)
    qv = (q + sd[prefix + ".v"]).reshape(B, Tg, H, dh).transpose(1, 2)
    k = k.reshape(B, Tkg, H, dh).transpose(1, 2)
    v = v.reshape(B, Tkg, H, dh).transpose(1, 2)
    e = linear(sd, prefix + ".pos_layer", grouped_pos_table(Tp, Tkp - Tp, D, G).to(x.dtype)).reshape(-1, H, dh).transpose(0, 1)     # (H, Tkg + Tg - 1, dh)
    s_k = qu @ k.transpose(2, 3)
    s_all = qv @ e.transpose(1, 2).unsqueeze(0)
    i = torch.arange(Tg).unsqueeze(1)
    j = torch.arange(Tkg).unsqueeze(0)
    idx = (Tg - 1) - i + j                                   # row r <-> group offset (Tkg - 1 - r); needed: (Tkg - Tg + i) - j
    s_e = s_all.gather(3, idx.expand(B, H, Tg, Tkg))
    scores = (s_k + s_e) / dh ** 0.5
    if mask is not None:
        scores = scores + mask[:, :, ::G, ::G].logical_not() * -1e9
    p = scores.softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Tp, D)[:, :T]
    o = linear(sd, prefix + ".output_layer", o)
    return (o, p, new_hidden) if return_hidden else o


def patch_attention(sd, prefix, x, mask, H, P):
    """nnet/attentions.py:348-382: pad to multiple of P, min-pool mask, avg-pool x (divisor P,
    zeros included), attention on ceil(T/P) tokens, nearest upsample xP, slice to T."""
    B, T, D = x.shape
    pad = (P - T % P) % P
    xp = F.pad(x, (0, 0, 0, pad))
    xp = xp.view(B, (T + pad) // P, P, D).mean(dim=2)
    mp = patch_pool_mask(mask, P) if mask is not None else None
    o = rel_pos_attention(sd, prefix, xp, mp, H)
    o = o.repeat_interleave(P, dim=1)
    return o[:, :T]


def feed_forward(sd, prefix, x):
    """nnet/modules.py:257-289 (dropout = identity here)."""
    h = layer_norm(sd, prefix + ".layers.0", x)
    h = swish(linear(sd, prefix + ".layers.1", h))
    return linear(sd, prefix + ".layers.4", h)


def conv_module(sd, prefix, x, stride, train, stats_out, causal=False):
    """nnet/modules.py:341-385: LN -> pw conv (D->2D') -> GLU -> depthwise k=15 stride s, zero pad
    (k//2, (k-1)//2) for "same" or (k-1, 0) for "causal" (nnet/layers.py:137-156) -> BatchNorm1d -> Swish -> pw conv."""
    h = layer_norm(sd, prefix + ".layers.0", x)
    h = F.linear(h, sd[prefix + ".layers.1.weight"][:, :, 0], sd[prefix + ".layers.1.bias"])
    h = F.glu(h, dim=-1)
    w = sd[prefix + ".layers.3.weight"]
    k = w.shape[-1]
    h = F.pad(h.transpose(1, 2), (k - 1, 0) if causal else (k // 2, (k - 1) // 2))
    h = F.conv1d(h, w, sd[prefix + ".layers.3.bias"], stride=stride, groups=w.shape[0])
    h = batch_norm(sd, prefix + ".layers.4", h, train, stats_out)
    h = swish(h).transpose(1, 2)
    return F.linear(h, sd[prefix + ".layers.6.weight"][:, :, 0], sd[prefix + ".layers.6.bias"])


def conformer_block(sd, prefix, x, mask, H, patch, train, stats_out, causal_conv=False, group=0):
    """nnet/blocks.py:289-306.  group > 0: GroupedRelPosMultiHeadSelfAttention with that group size (Transformer-XL biases), else patch / plain rel-pos attention."""
    x = x + 0.5 * feed_forward(sd, prefix + ".ff_module1", x)
    h = layer_norm(sd, prefix + ".self_att_module.norm", x)
    if group > 0:
        a = grouped_rel_pos_attention(sd, prefix + ".self_att_module.attention", h, mask, H, group)
    elif patch > 1:
        a = patch_attention(sd, prefix + ".self_att_module.attention", h, mask, H, patch)
    else:
        a = rel_pos_attention(sd, prefix + ".self_att_module.attention", h, mask, H)
    x = x + a
    stride = 1
    if prefix + ".conv_res.weight" in sd:
        stride = 2
        res = F.linear(x[:, ::2], sd[prefix + ".conv_res.weight"][:, :, 0], sd[prefix + ".conv_res.bias"])
    else:
        res = x
    x = res + conv_module(sd, prefix + ".conv_module", x, stride, train, stats_out, causal_conv)
    x = x + 0.5 * feed_forward(sd, prefix + ".ff_module2", x)
    return layer_norm(sd, prefix + ".norm", x), stride


def conformer_interctc(sd, prefix, x, lengths, num_blocks, interctc_blocks, loss_prefix, patch_sizes,
                       train, stats_out, H=4, context=None, causal_conv=False, group_sizes=None):
    """nnet/networks.py:262-307.  context = (left_context, right_context, mask_start) of a streaming Mask, or None for the key-padding mask.
    group_sizes: per stage, the group size of GroupedRelPosMultiHeadSelfAttention (att_type="grouped", nnet/networks.py:389-392) instead of patch / plain attention."""
    T = x.shape[1]
    if context is not None:
        mask = context_mask(T, lengths, *context)
    else:
        mask = key_padding_mask(T, lengths) if lengths is not None else None
    inter = {}
    i, j = 0, 0
    for stage, nb in enumerate(num_blocks):
        for _ in range(nb):
            x, stride = conformer_block(sd, f"{prefix}.conformer_blocks.{i}", x, mask, H,
                                        patch_sizes[stage], train, stats_out, causal_conv, group=group_sizes[stage] if group_sizes else 0)
            logits = None
            if i + 1 in interctc_blocks:
                p = f"{prefix}.interctc_modules.{j}"
                logits = linear(sd, p + ".proj_1", x)
                x = x + linear(sd, p + ".proj_2", logits.softmax(dim=-1))
                j += 1
            if stride > 1:
                if mask is not None:
                    mask = mask[:, :, ::stride, ::stride]
                if lengths is not None:
                    lengths = strided_lengths(lengths, stride)
            if logits is not None:
                inter[f"{loss_prefix}_{i}"] = [logits, lengths]
            i += 1
    return x, lengths, inter


# --------------------------------------------------------------------------------------------
# front-ends
# --------------------------------------------------------------------------------------------


def audio_stem(sd, prefix, mel, lengths, train, stats_out):
    """nnet/networks.py:356-377,419-432 + nnet/modules.py:70-130: Conv2d(1->180,3x3,s2,'same' =
    explicit zero pad (1,1,1,1)) -> BN2d -> Swish; (B,180,40,T') -> (B,T',7200) -> Linear."""
    x = F.pad(mel[:, None], (1, 1, 1, 1))
    x = F.conv2d(x, sd[prefix + ".subsampling_module.layers.0.0.weight"],
                 sd[prefix + ".subsampling_module.layers.0.0.bias"], stride=2)
    x = swish(batch_norm(sd, prefix + ".subsampling_module.layers.0.1", x, train, stats_out))
    lengths = strided_lengths(lengths, 2)
    B, C, Fq, T = x.shape
    x = x.reshape(B, C * Fq, T).transpose(1, 2)
    return linear(sd, prefix + ".linear", x), lengths


def resnet_block(sd, prefix, x, stride, train, stats_out):
    """nnet/blocks.py:29-91 (basic block, joined post-activation)."""
    h = F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[prefix + ".layers.0.weight"], None, stride=stride)
    h = F.relu(batch_norm(sd, prefix + ".layers.1", h, train, stats_out))
    h = F.conv2d(F.pad(h, (1, 1, 1, 1)), sd[prefix + ".layers.3.weight"], None)
    h = batch_norm(sd, prefix + ".layers.4", h, train, stats_out)
    if prefix + ".residual.0.weight" in sd:
        r = F.conv2d(x, sd[prefix + ".residual.0.weight"], None, stride=stride)
        r = batch_norm(sd, prefix + ".residual.1", r, train, stats_out)
    else:
        r = x
    return F.relu(h + r)


def visual_frontend(sd, prefix, video, train, stats_out):
    """nnet/networks.py:459-473,497-504: Conv3d(1->64,(5,7,7),s(1,2,2), zero pad (2,2)(3,3)(3,3), bias)
    -> BN3d -> ReLU -> MaxPool3d((1,3,3),s(1,2,2), zero pad 1) -> frames -> ResNet-18 -> mean 3x3
    -> Linear(512->256).  video: (B,1,T,H,W)."""
    x = F.pad(video, (3, 3, 3, 3, 2, 2))
    x = F.conv3d(x, sd[prefix + ".0.layers.0.0.weight"], sd[prefix + ".0.layers.0.0.bias"], stride=(1, 2, 2))
    x = F.relu(batch_norm(sd, prefix + ".0.layers.0.1", x, train, stats_out))
    x = F.max_pool3d(F.pad(x, (1, 1, 1, 1, 0, 0)), (1, 3, 3), (1, 2, 2))
    B, C, T, Hh, Ww = x.shape
    x = x.transpose(1, 2).reshape(B * T, C, Hh, Ww)
    for bi in range(8):
        stride = 2 if (bi % 2 == 0 and bi > 0) else 1
        x = resnet_block(sd, f"{prefix}.3.blocks.{bi}", x, stride, train, stats_out)
    x = x.mean(dim=(2, 3))
    x = linear(sd, prefix + ".3.head.1", x)
    return x.view(B, T, -1)


# --------------------------------------------------------------------------------------------
# models (a16) and loss (a17)
# --------------------------------------------------------------------------------------------


def av_forward(sd, video, video_len, audio, audio_len, train=True, stats_out=None,
               v_interctc=(3, 6), a_interctc=(8, 11), f_interctc=(2,)):
    """nnet/models_zoo.py:156-161 + nnet/networks.py:559-579.  video: (B,T,H,W,1)."""
    p = "encoder"
    v = visual_frontend(sd, p + ".video_encoder.front_end", video.permute(0, 4, 1, 2, 3), train, stats_out)
    v, vlen, v_inter = conformer_interctc(sd, p + ".video_encoder.back_end", v, video_len, [6, 1],
                                          v_interctc, "v_ctc", [1, 1], train, stats_out)
    mel, alen = mel_frontend(audio, audio_len)
    a, alen = audio_stem(sd, p + ".audio_encoder", mel, alen, train, stats_out)
    a, alen, a_inter = conformer_interctc(sd, p + ".audio_encoder.back_end", a, alen, [5, 6, 1],
                                          a_interctc, "a_ctc", [3, 1, 1], train, stats_out)
    x = torch.cat([a, v], dim=-1)
    x = linear(sd, p + ".fusion_module.layers.2", swish(linear(sd, p + ".fusion_module.layers.0", x)))
    x, lengths, f_inter = conformer_interctc(sd, p + ".audio_visual_encoder", x, alen, [5], f_interctc,
                                             "f_ctc", [1], train, stats_out)
    x = linear(sd, p + ".head", x)
    out = {"outputs": [x, lengths]}
    out.update(f_inter)
    out.update(v_inter)
    out.update(a_inter)
    return out


def vo_forward(sd, video, video_len, train=True, stats_out=None, interctc=(3, 6, 9)):
    """nnet/models_zoo.py:99-147 + nnet/networks.py:442-512 (VisualEfficientConformerInterCTC): video (B,T,H,W,1) -> {"outputs", "ctc_2", "ctc_5", "ctc_8"}."""
    p = "encoder"
    v = visual_frontend(sd, p + ".front_end", video.permute(0, 4, 1, 2, 3), train, stats_out)
    v, vlen, inter = conformer_interctc(sd, p + ".back_end", v, video_len, [6, 6], interctc, "ctc", [1, 1], train, stats_out)
    out = {"outputs": [linear(sd, p + ".head", v), vlen]}
    out.update(inter)
    return out


VO_LOSS_WEIGHTS = [0.5 / 3, 0.5 / 3, 0.5 / 3, 0.5]      # list, mapped positionally onto (outputs, ctc_2, ctc_5, ctc_8): nnet/models_zoo.py:124, nnet/model.py:119-149


def lrw_forward(sd, video, train=True, stats_out=None):
    """nnet/models_zoo.py:33-41 (VisualEfficientConformerCE): visual encoder (nnet/networks.py:442-512, num_blocks [6,6], no InterCTC, lengths None = no
    mask) -> head -> mean over time.  video: (B,1,T,H,W) -> logits (B, vocab)."""
    p = "encoder"
    v = visual_frontend(sd, p + ".front_end", video, train, stats_out)
    v, _, _ = conformer_interctc(sd, p + ".back_end", v, None, [6, 6], (), "ctc", [1, 1], train, stats_out)
    return linear(sd, p + ".head", v).mean(dim=1)


def softmax_cross_entropy(logits, targets, ignore_index=-1):
    """nnet/losses.py:258-290: per-element cross entropy (0 where the target is ignore_index), then the mean over ALL elements."""
    lp = F.log_softmax(logits.float(), dim=-1)
    keep = targets != ignore_index
    nll = -lp.gather(-1, targets.clamp(min=0).unsqueeze(-1)).squeeze(-1)
    return torch.where(keep, nll, torch.zeros_like(nll)).mean()


def ao_forward(sd, audio, audio_len, train=True, stats_out=None, interctc=(3, 6, 10, 13), num_blocks=(5, 6, 5), att_type="patch"):
    """nnet/models_zoo.py:64-97 + nnet/networks.py:411-440 (att_type 'patch', or 'grouped': group sizes 3, 1, 1)."""
    p = "encoder"
    mel, alen = mel_frontend(audio, audio_len)
    a, alen = audio_stem(sd, p, mel, alen, train, stats_out)
    a, alen, inter = conformer_interctc(sd, p + ".back_end", a, alen, list(num_blocks), interctc, "ctc",
                                        [3, 1, 1] if att_type == "patch" else [1, 1, 1], train, stats_out, group_sizes=[3, 1, 1] if att_type == "grouped" else None)
    x = linear(sd, p + ".head", a)
    out = {"outputs": [x, alen]}
    out.update(inter)
    return out


def ctc_nll(logits, logit_len, targets, target_len, blank=0):
    """nnet/losses.py:311-334: per-utterance -log p(y|x) from log_softmax(logits), summed over frames
    (reduction='none'), zero_infinity=True.  Own log-space alpha recursion (not aten::_ctc_loss)."""
    B, T, V = logits.shape
    lp = F.log_softmax(logits.float(), dim=-1)
    out = []
    neg_inf = float("-inf")
    for b in range(B):
        Tb, L = int(logit_len[b]), int(target_len[b])
        y = targets[b, :L].tolist()
        ext = [blank]
        for t in y:
            ext += [t, blank]
        S = len(ext)
        ext_t = torch.tensor(ext)
        alpha = torch.full((S,), neg_inf, dtype=lp.dtype)
        if Tb > 0:
            alpha[0] = lp[b, 0, blank]
            if S > 1:
                alpha[1] = lp[b, 0, ext[1]]
        can_skip = torch.zeros(S, dtype=torch.bool)
        for s in range(2, S):
            can_skip[s] = ext[s] != blank and ext[s] != ext[s - 2]
        for t in range(1, Tb):
            a1 = torch.cat([alpha.new_full((1,), neg_inf), alpha[:-1]])[:S]
            a2 = torch.cat([alpha.new_full((2,), neg_inf), alpha[:-2]])[:S]
            a2 = torch.where(can_skip, a2, alpha.new_full((S,), neg_inf))
            alpha = torch.logsumexp(torch.stack([alpha, a1, a2]), dim=0) + lp[b, t, ext_t]
        if Tb == 0:
            ll = torch.tensor(0.0 if L == 0 else neg_inf)
        elif S > 1:
            ll = torch.logsumexp(torch.stack([alpha[S - 1], alpha[S - 2]]), dim=0)
        else:
            ll = alpha[S - 1]
        nll = -ll
        nll = torch.where(torch.isinf(nll), torch.zeros_like(nll), nll)  # zero_infinity
        out.append(nll)
    return torch.stack(out)


AV_LOSS_WEIGHTS = {"v_ctc_2": 0.5 / 3, "v_ctc_5": 0.5 / 3, "a_ctc_7": 0.5 / 3, "a_ctc_10": 0.5 / 3,
                   "f_ctc_1": 0.5 / 3, "outputs": 0.5}


def total_loss(outputs, targets, target_len, weights, use_aten=True):
    """nnet/model.py:275-287: sum_k w_k * mean_b CTC_k."""
    losses = {}
    total = 0.0
    for key, (logits, lens) in outputs.items():
        if use_aten:
            nll = F.ctc_loss(F.log_softmax(logits if logits.dtype == torch.float64 else logits.float(), dim=-1).transpose(0, 1), targets, lens, target_len,
                             blank=0, reduction="none", zero_infinity=True)
        else:
            nll = ctc_nll(logits, lens, targets, target_len)
        losses["loss_" + key] = nll.mean()
        total = total + losses["loss_" + key] * weights[key]
    losses["loss"] = total
    return losses


# --------------------------------------------------------------------------------------------
# optimizer (a18)
# --------------------------------------------------------------------------------------------


def noam_lr(step, warmup=10000, dim=360, factor=2.0):
    """nnet/schedulers.py:120-137."""
    return factor * dim ** -0.5 * min(step * warmup ** -1.5, step ** -0.5)


def adam_update(param, grad, m, v, step, lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6):
    """torch.optim.Adam semantics used by nnet/optimizers.py:61-93 (coupled L2 weight decay)."""
    b1, b2 = betas
    g = grad + weight_decay * param
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    denom = (v.sqrt() / math.sqrt(1 - b2 ** step)) + eps
    return param - lr * mhat / denom, m, v
