"""Encoders (nnet/networks.py): ResNet-18 visual front-end, ConformerInterCTC stack, audio / visual / audio-visual encoders."""
import os

import torch
import torch.nn as nn

from .. import ops
from .. import runtime as rt
from . import attentions, blocks, layers, modules, normalizations, preprocessing, transforms


class ResNet(nn.Module):
    """nnet/networks.py:32-146.  Hot path = ResNet18 without stem; forward_nhwc consumes the stem's channels-last frames."""

    def __init__(self, dim_input=3, dim_output=1000, model="ResNet50", include_stem=True, include_head=True):
        super().__init__()
        assert model in ("ResNet18", "ResNet34") and not include_stem, "hot path: basic-block ResNet fed by the Conv3d stem (nnet/networks.py:472)"
        dim_stem, dim_blocks = 64, [64, 128, 256, 512]
        num_blocks = [2, 2, 2, 2] if model == "ResNet18" else [3, 4, 6, 3]
        self.stem = nn.Identity()
        self.blocks = nn.ModuleList()
        for stage in range(4):
            for b in range(num_blocks[stage]):
                first = b == 0
                cin = (dim_stem if stage == 0 else dim_blocks[stage - 1]) if first else dim_blocks[stage]
                stride = (2, 2) if (first and stage > 0) else (1, 1)
                self.blocks.append(blocks.ResNetBlock(in_features=cin, out_features=dim_blocks[stage], kernel_size=(3, 3), stride=stride,
                                                      act_fun="ReLU", joined_post_act=True))
        self.head = nn.Sequential(layers.GlobalAvgPool2d(),
                                  layers.Linear(dim_blocks[-1], dim_output, weight_init="he_normal", bias_init="zeros")) if include_head else nn.Identity()

    def forward_nhwc(self, x):
        for i, blk in enumerate(self.blocks):
            x = blk.forward_nhwc(x, chain=i + 1 < len(self.blocks))       # every block but the last hands its output to the next block only
        if isinstance(self.head, nn.Identity):
            return x
        x = ops.AvgPoolFn.apply(x)
        lin = self.head[1]
        return ops.linear(x, lin.weight, lin.bias)

    def forward(self, x):
        return self.forward_nhwc(x.permute(0, 2, 3, 1).to(rt.act_dtype()).contiguous())


class ConformerInterCTC(nn.Module):
    """nnet/networks.py:202-307"""

    def __init__(self, dim_model, num_blocks, interctc_blocks, vocab_size, loss_prefix="ctc", att_params={"class": "MultiHeadAttention", "num_heads": 4},
                 conv_params={"class": "Conv1d", "params": {"padding": "same", "kernel_size": 31}}, ff_ratio=4, drop_rate=0.1, pos_embedding=None,
                 mask=None, conv_stride=1, batch_norm=True):
        super().__init__()
        assert pos_embedding is None
        self.interctc_blocks, self.loss_prefix = interctc_blocks, loss_prefix
        dim_model = [dim_model] if isinstance(dim_model, int) else dim_model
        num_blocks = [num_blocks] if isinstance(num_blocks, int) else num_blocks
        self.pos_embedding = None
        self.dropout = layers.Dropout(p=drop_rate)
        self.mask = mask
        self.conformer_blocks = nn.ModuleList()
        self.interctc_modules = nn.ModuleList()
        i = 1
        for stage, nb in enumerate(num_blocks):
            for b in range(nb):
                down = (b == nb - 1) and (stage < len(num_blocks) - 1)       # last block of every stage but the last: stride + widen
                d_out = dim_model[stage + 1] if down else dim_model[stage]
                stride = (conv_stride[stage] if isinstance(conv_stride, list) else conv_stride) if down else 1
                self.conformer_blocks.append(blocks.ConformerBlock(
                    dim_model=dim_model[stage], dim_expand=d_out, ff_ratio=ff_ratio, drop_rate=drop_rate,
                    att_params=att_params[stage] if isinstance(att_params, list) else att_params, conv_stride=stride,
                    conv_params=conv_params[stage] if isinstance(conv_params, list) else conv_params, batch_norm=batch_norm))
                if i in interctc_blocks:
                    self.interctc_modules.append(modules.InterCTCResModule(dim_model=d_out, vocab_size=vocab_size))
                i += 1

        # the LayerNorm that closes block i is read next by the first pre-norm of block i + 1 (unless an InterCTC module sits between them): one launch for both
        # (ops.LN_PAIR); a plain attribute, not a registered submodule (the state_dict is the reference's)
        for i in range(len(self.conformer_blocks) - 1):
            b0, b1 = self.conformer_blocks[i], self.conformer_blocks[i + 1]
            if (i + 1) not in interctc_blocks and isinstance(b0.norm, nn.LayerNorm) and b0.norm.normalized_shape == b1.ff_module1.layers[0].normalized_shape:
                object.__setattr__(b0, "_next_ln", b1.ff_module1.layers[0])
        # the position projections of the consecutive blocks of a stage share their input (the sinusoid table of the stage's sequence length): one [L D][D] weight in the
        # arena -> one launch per stage and pass instead of one per block (ops._pos_group_entry); state_dict names are unchanged
        i0 = 0
        for nb in num_blocks:
            atts = [getattr(b.self_att_module, "attention", None) for b in self.conformer_blocks[i0:i0 + nb]]
            i0 += nb
            pos = [a.pos_layer for a in atts if a is not None and type(a).__name__ in ("RelPos1dMultiHeadAttention", "RelPosPatch1dMultiHeadAttention") and hasattr(a, "pos_layer")]
            if len(pos) == len(atts) and len(pos) >= 2 and all(p.bias is not None and p.weight.shape == pos[0].weight.shape for p in pos):
                rt.fuse_linears(self, [p.weight for p in pos], [p.bias for p in pos])
        if any(getattr(m, "causal", False) for m in self.modules()):
            # causal relative positions (nnet/attentions.py:234-256): for j <= i the reference reads E[i-j], as the full-context layout does; for j > i its
            # rel_to_abs wraps around into the next query row.  Only a mask that hides every j > i makes that well defined -- required here, not reproduced.
            ok = mask is not None and getattr(mask, "right_context", None) is not None and mask.right_context <= 0 and getattr(mask, "mask_start", 0) <= 1
            assert ok, "causal=True attention needs a causal Mask (right_context <= 0, mask_start <= 1)"

    def forward(self, x, lengths):
        x = self.dropout(x)
        if self.mask is not None and getattr(self.mask, "has_context", False):
            mask = self.mask(x, lengths)              # streaming: dense (B or 1,1,T,T) band mask, strided with the blocks (nnet/networks.py:271-298)
        else:
            mask = modules.LengthMask(lengths) if (self.mask is not None and lengths is not None) else None
        inter, j = {}, 0
        for i, block in enumerate(self.conformer_blocks):
            x = block(x, mask=mask)
            logits = None
            if i + 1 in self.interctc_blocks:
                x, logits = self.interctc_modules[j](x)
                j += 1
            if block.stride > 1:
                same = isinstance(mask, modules.LengthMask) and mask.lengths is lengths      # the length mask carries the very vector that is strided below: computed once
                if lengths is not None:
                    lengths = ops.len_affine(lengths, 1, block.stride, 1)
                if mask is not None:
                    if isinstance(mask, modules.LengthMask):
                        mask = modules.LengthMask(lengths) if same else mask.strided(block.stride)
                    else:
                        mask = mask[:, :, ::block.stride, ::block.stride]
            if logits is not None:
                inter[self.loss_prefix + "_" + str(i)] = [logits, lengths]
        return x, lengths, inter


def _relpos(num_heads, attn_drop_rate, max_pos, **extra):
    cls = "RelPosPatch1dMultiHeadAttention" if "patch_size" in extra else "RelPos1dMultiHeadAttention"
    return {"class": cls, "params": dict(num_heads=num_heads, attn_drop_rate=attn_drop_rate, num_pos_embeddings=max_pos,
                                         weight_init="default", bias_init="default", **extra)}


class AudioEfficientConformerEncoder(nn.Module):
    """nnet/networks.py:309-440"""

    def __init__(self, include_head=True, vocab_size=256, att_type="patch", interctc_blocks=[3, 6, 10, 13], num_blocks=[5, 6, 5], loss_prefix="ctc"):
        super().__init__()
        assert att_type in ("regular", "grouped", "patch")
        filters, n_mels, dim_model, H = 180, 80, [180, 256, 360], 4
        self.audio_preprocessing = preprocessing.AudioPreprocessing(16000, 512, 25, 10, n_mels, False, -5.6501, 4.2280)
        self.spec_augment = preprocessing.SpecAugment(mF=2, F=27, mT=5, pS=0.05)
        self.unsqueeze = layers.Unsqueeze(dim=1)
        self.subsampling_module = modules.ConvNeuralNetwork(dim_input=1, dim_layers=filters, kernel_size=3, strides=2, norm="BatchNorm2d", act_fun="Swish",
                                                            drop_rate=0.0, dim=2)
        self.reshape = layers.Reshape(shape=(filters * n_mels // 2, -1), include_batch=False)
        self.transpose = layers.Transpose(1, 2)
        self.linear = layers.Linear(filters * n_mels // 2, dim_model[0])
        if att_type == "grouped":             # nnet/networks.py:389-392: Transformer-XL biases everywhere, 3-frame groups in the first stage
            grouped = lambda G: {"class": "GroupedRelPosMultiHeadSelfAttention", "params": {"num_heads": H, "group_size": G, "attn_drop_rate": 0.0, "max_pos_encoding": 10000, "causal": False}}
            att = [grouped(3), grouped(1), grouped(1)]
        else:
            att = [_relpos(H, 0.0, 10000, patch_size=3) if att_type == "patch" else _relpos(H, 0.0, 10000), _relpos(H, 0.0, 10000), _relpos(H, 0.0, 10000)]
        self.back_end = ConformerInterCTC(dim_model=dim_model, num_blocks=num_blocks, interctc_blocks=interctc_blocks, vocab_size=vocab_size,
                                          att_params=att,
                                          conv_params={"class": "Conv1d", "params": {"padding": "same", "kernel_size": 15}}, ff_ratio=4, drop_rate=0.1,
                                          pos_embedding=None, mask=attentions.Mask(), conv_stride=2, batch_norm=True, loss_prefix=loss_prefix)
        self.head = layers.Linear(dim_model[-1], vocab_size) if include_head else nn.Identity()

    def forward(self, x, lengths):
        mel, lengths = self.audio_preprocessing(x, lengths)
        mel = self.spec_augment(mel, lengths)
        stem = self.subsampling_module.layers[0]
        a = ops.AudioStemFn.apply(mel, stem[0].weight, stem[0], stem[1], stem[1].training and not stem[1].frozen)     # (B, T', 7200) act
        lengths = ops.len_affine(lengths, 1, 2, 1)
        x = ops.linear(a, self.linear.weight, self.linear.bias)
        x, lengths, inter = self.back_end(x, lengths)
        if not isinstance(self.head, nn.Identity):
            x = self.head(x)
        return x, lengths, inter


class VisualEfficientConformerEncoder(nn.Module):
    """nnet/networks.py:442-512"""

    def __init__(self, include_head=True, vocab_size=256, interctc_blocks=[3, 6, 9], num_blocks=[6, 6], loss_prefix="ctc"):
        super().__init__()
        dim_model, H = [256, 360], 4
        self.front_end = nn.Sequential(
            modules.ConvNeuralNetwork(dim_input=1, dim_layers=64, kernel_size=(5, 7, 7), strides=(1, 2, 2), norm="BatchNorm3d", act_fun="ReLU", drop_rate=0.0, dim=3),
            layers.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding="same"),
            transforms.VideoToImages(),
            ResNet(include_stem=False, dim_output=dim_model[0], model="ResNet18"))
        self.expand_time = transforms.ImagesToVideos()
        self.back_end = ConformerInterCTC(dim_model=dim_model, num_blocks=num_blocks, interctc_blocks=interctc_blocks, vocab_size=vocab_size,
                                          att_params=_relpos(H, 0.0, 10000), conv_params={"class": "Conv1d", "params": {"padding": "same", "kernel_size": 15}},
                                          ff_ratio=4, drop_rate=0.1, pos_embedding=None, mask=attentions.Mask(), conv_stride=2, batch_norm=True,
                                          loss_prefix=loss_prefix)
        self.head = layers.Linear(dim_model[-1], vocab_size) if include_head else nn.Identity()

    def forward_front(self, x):
        """x: (B, 1, T, H, W) -> per-frame features (B, T, 256) fp32: Conv3d stem + max-pool + ResNet-18 (nnet/networks.py:497-504)"""
        B, C, T, H, W = x.shape
        assert C == 1
        stem = self.front_end[0].layers[0]
        bn = stem[1]
        frames = ops.VideoStemFn.apply(x.reshape(B, T, H, W), stem[0].weight, stem[0], bn, bn.training and not bn.frozen)     # (B*T, H/4, W/4, 64) act, channels-last
        frames = ops.mark(frames, "v_stem")
        feats = self.front_end[3].forward_nhwc(frames)                                                          # (B*T, 256) fp32
        return feats.view(B, T, -1)

    def forward_back(self, x, lengths):
        x, lengths, inter = self.back_end(x, lengths)
        if not isinstance(self.head, nn.Identity):
            x = self.head(x)
        return x, lengths, inter

    def forward(self, x, lengths):
        """x: (B, 1, T, H, W)"""
        return self.forward_back(self.forward_front(x), lengths)


def audio_first_env():
    return os.environ.get("AVEC_AUDIO_FIRST", "0") == "1"


class AudioVisualEfficientConformerEncoder(nn.Module):
    """nnet/networks.py:514-579"""

    def __init__(self, include_head=True, vocab_size=256, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2]):
        super().__init__()
        dim_model = 360
        self.video_encoder = VisualEfficientConformerEncoder(include_head=False, vocab_size=vocab_size, interctc_blocks=v_interctc_blocks,
                                                             num_blocks=[6, 1], loss_prefix="v_ctc")
        self.audio_encoder = AudioEfficientConformerEncoder(include_head=False, vocab_size=vocab_size, interctc_blocks=a_interctc_blocks,
                                                            num_blocks=[5, 6, 1], loss_prefix="a_ctc")
        self.fusion_module = modules.FusionModule(a_dim_model=dim_model, v_dim_model=dim_model, f_dim_model=dim_model)
        self.audio_visual_encoder = ConformerInterCTC(dim_model=dim_model, num_blocks=5, interctc_blocks=f_interctc_blocks, vocab_size=vocab_size,
                                                      att_params=_relpos(4, 0.0, 10000),
                                                      conv_params={"class": "Conv1d", "params": {"padding": "same", "kernel_size": 15}}, ff_ratio=4,
                                                      drop_rate=0.1, pos_embedding=None, mask=attentions.Mask(), conv_stride=2, batch_norm=True, loss_prefix="f_ctc")
        self.head = layers.Linear(dim_model, vocab_size) if include_head else nn.Identity()

    def forward(self, video, video_len, audio, audio_len):
        side = rt.branch_stream() if (video.is_cuda and audio.is_cuda) else None
        if side is None:
            video, video_len, v_inter = self.video_encoder(video, video_len)
            audio, audio_len, a_inter = self.audio_encoder(audio, audio_len)
        else:
            # the two encoders are independent: the audio branch runs on a second stream beside the visual one (runtime.branch_stream)
            main = torch.cuda.current_stream()
            rt.stream()                                   # default reduction workspace registered before the fork
            # weight shadows: the visual front-end's on this stream (the stem starts at once), the other 80 % on the audio stream beside the stem's forward
            # pass (a persistent kernel that leaves the HBM idle); this stream waits for them in front of the visual back-end.
            arena = rt.arena_of(self)
            sh_ev = None
            split = arena is not None and not audio_first_env()
            if split:
                side.wait_stream(main)
                sh_ev = arena.ensure_fresh_split(arena.prefix_blocks(self.video_encoder.front_end), side)
            else:
                rt.ensure_shadows_fresh(self)
                side.wait_stream(main)
            for t in (audio, audio_len):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(side)
            # Host launch order = order in which a captured graph's kernels reach the GPU.  Forward: the visual front-end (few long kernels, the critical
            # path) goes first, the ~400 short audio kernels are submitted while it runs, the visual conformer stack follows.  The autograd engine replays
            # nodes newest-first, so the backward order is: visual conformer stack, audio encoder, ResNet / stem -- the audio submissions again hide
            # behind running work instead of delaying the critical branch.  AVEC_AUDIO_FIRST=1 restores the old order (audio encoder first).
            audio_first = audio_first_env()
            if audio_first:
                with torch.cuda.stream(side):
                    audio, audio_len, a_inter = self.audio_encoder(audio, audio_len)
                video, video_len, v_inter = self.video_encoder(video, video_len)
            else:
                ops.stamp("step_start:f")
                feats = ops.mark(self.video_encoder.forward_front(video), "v_front")
                with torch.cuda.stream(side):
                    ops.stamp("a_start:f")
                    audio, audio_len, a_inter = self.audio_encoder(audio, audio_len)
                    audio = ops.mark(audio, "a_enc")
                if sh_ev is not None:
                    main.wait_event(sh_ev)
                video, video_len, v_inter = self.video_encoder.forward_back(feats, video_len)
                video = ops.mark(video, "v_back")
            main.wait_stream(side)
            for t in [audio, audio_len] + [u for v in a_inter.values() for u in (v if isinstance(v, (list, tuple)) else [v])]:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(main)
        arena = rt.arena_of(self.fusion_module) if audio.is_cuda else None
        if arena is not None and getattr(arena, "_early_armed", False):
            # distributed training: when the gradient reaches these two tensors, the fusion module / audio-visual encoder / head gradients are final ->
            # their all-reduce starts while the two encoders are still back-propagating; the audio encoder's follows when its own backward ends
            r_f, r_h = arena.range_of(self.fusion_module), arena.range_of(self.head) if not isinstance(self.head, nn.Identity) else arena.range_of(self.audio_visual_encoder)
            r_a = arena.range_of(self.audio_encoder)
            arena._boundary_fired = False
            arena._audio_range = r_a
            if r_f is not None and r_h is not None and r_h[1] >= r_f[0]:
                audio, video = rt.grad_boundary(audio, arena, r_f[0], r_h[1]), rt.grad_boundary(video, arena, r_f[0], r_h[1])
        x = ops.mark(self.fusion_module(audio, video), "fusion")
        x, lengths, inter = self.audio_visual_encoder(x, audio_len)
        x = ops.mark(x, "av_enc")
        inter.update(v_inter)
        inter.update(a_inter)
        if not isinstance(self.head, nn.Identity):
            x = self.head(x)
        return x, lengths, inter
