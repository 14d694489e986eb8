"""Report (GPU box): per-tensor relative L2 error of the bf16 HIP gradients against the fp64 oracle, next to the fp32 oracle's and fp32 HIP path's own errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import bf16_grad_probe as probe
from oracle import avec_oracle as O


def oracle_grads(sd0, dtype):
    video, vlen, audio, alen, labels, llen = probe.av_inputs(2)
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    out = O.av_forward(sd, video.to(dtype), vlen, audio.to(dtype), alen, train=True, stats_out={})
    O.total_loss(out, labels, llen, O.AV_LOSS_WEIGHTS)["loss"].backward()
    return {k: v.grad.clone() for k, v in sd.items() if v.requires_grad and v.grad is not None}


model, sd0 = probe.build_model()
g64, g32 = oracle_grads(sd0, torch.float64), oracle_grads(sd0, torch.float32)
e16, _, _ = probe.grad_errors(model, g64, "bf16")
model.load_state_dict(sd0)
e32, _, _ = probe.grad_errors(model, g64, "f32")
SZ = ("key_layer.bias", "pos_layer.bias", "conv_module.layers.3.bias", "layers.0.0.bias")
rows = sorted(((e16[k], k) for k in e16 if not k.endswith(SZ)), reverse=True)
l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
print("n=%d median=%.4f p90=%.4f p99=%.4f" % (len(rows), rows[len(rows) // 2][0], rows[len(rows) // 10][0], rows[len(rows) // 100][0]))
for e, k in rows[:40]:
    print("%.4f  hip_f32 %.4f  oracle_f32 %.5f  |g64| %.3e  n=%d  %s" % (e, e32[k], l2(g32[k], g64[k]), g64[k].norm().item(), g64[k].numel(), k))
import collections, re
grp = collections.defaultdict(list)
for e, k in rows:
    m = re.match(r"encoder\.(video_encoder\.front_end\.0|video_encoder\.front_end\.3\.blocks\.\d+|video_encoder\.front_end\.3\.head|video_encoder\.back_end\.conformer_blocks\.\d+|video_encoder\.back_end\.interctc_modules\.\d+|audio_encoder\.back_end\.conformer_blocks\.\d+|audio_encoder\.[a-z_]+|fusion_module|audio_visual_encoder\.conformer_blocks\.\d+|audio_visual_encoder\.interctc_modules\.\d+|head)", k)
    grp[m.group(1) if m else k.split(".")[1]].append((e, e32[k]))
print("--- groups: median / max bf16 error, median fp32-HIP error")
for gname in sorted(grp, key=lambda s: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]):
    v = sorted(grp[gname])
    print("%-60s n=%3d  bf16 med %.4f max %.4f   f32 med %.4f" % (gname, len(v), v[len(v) // 2][0], v[-1][0], sorted(x[1] for x in v)[len(v) // 2]))
