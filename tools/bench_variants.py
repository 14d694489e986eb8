"""Secondary workloads of SURVEY.md 8(d) on one MI355X (not the bench.py line: these are the other BASELINE configurations and the
variable-length mixes, reported as utterances/s and padded-frame efficiency):

    python tools/bench_variants.py [--only NAME] [--steps K]

  av15s      AV model, 15 s utterances (audio 240 000 samples = 1501 mel frames, video 376 frames), B=8
  lrs2_main  AV model, B=32, durations ~ clipped log-normal(median 2.0 s, sigma 0.6) in [0.8, 6.2] s, zero-padded to the batch maximum
  lrs2_main_bucketed  the same distribution, 512 clips batched by the length-bucketed sampler (nnet/samplers.py)
  lrs2_pre   AV model, B=16, median 6 s, cap 16 s (pre-train mix)
  ao         audio-only InterCTC model, B=32, 63 840 samples
  vo         visual-only InterCTC model, B=32, 100 frames
  lrw        LRW word classifier (VisualEfficientConformerCE, 500 classes), B=64, 29 frames, cross-entropy

Steps run eagerly (variable shapes), bf16; one JSON line per workload."""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def durations(B, median, sigma, lo, hi, g):
    d = torch.exp(math.log(median) + sigma * torch.randn(B, generator=g))
    return d.clamp(lo, hi)


def av_batch(B, dur, g, device):
    alen = (16000 * dur).floor().long()
    vlen = alen // 640 + 1
    Ta, Tv = int(alen.max()), int(vlen.max())
    audio, video = torch.zeros(B, Ta), torch.zeros(B, Tv, 88, 88, 1)
    llen = (2.4 * dur).ceil().long()
    labels = torch.zeros(B, int(llen.max()), dtype=torch.long)
    for b in range(B):
        audio[b, :alen[b]] = 0.1 * torch.randn(int(alen[b]), generator=g)
        video[b, :vlen[b]] = torch.randn(int(vlen[b]), 88, 88, 1, generator=g)
        labels[b, :llen[b]] = torch.randint(1, 256, (int(llen[b]),), generator=g)
    eff = float(alen.sum()) / (B * Ta)
    return [video.to(device), vlen.to(device), audio.to(device), alen.to(device)], (labels.to(device), llen.to(device)), eff, B


GRAPHS = {"on": False, "bucket": None}


def run(name, model, batches, steps, warmup, precision):
    import avec_amd
    if GRAPHS["on"]:
        one = lambda inp, tgt: model.graphed_train_step(inp, tgt, precision=precision, bucket_frames=GRAPHS["bucket"], cache_size=max(8, len(batches)))
        name += "+graphs" + ("(bucket %d)" % GRAPHS["bucket"] if GRAPHS["bucket"] else "")
    else:
        one = lambda inp, tgt: model.train_step(inp, tgt, precision=precision)[0]
    warmup = max(warmup, len(batches))                  # every shape visited (captured) before the timed region: first visits pay for tables, allocator growth, captures
    for i in range(warmup):
        inp, tgt = batches[i % len(batches)][:2]
        last = one(inp, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        inp, tgt = batches[i % len(batches)][:2]
        last = one(inp, tgt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = float(last["loss"].detach())
    assert loss == loss and abs(loss) < 1e7, (name, loss)
    utt = sum(batches[i % len(batches)][3] for i in range(steps))
    return {"workload": name, "utt_per_s": round(utt / dt, 1), "ms_per_step": round(1e3 * dt / steps, 2), "steps": steps,
            "padded_frame_efficiency": round(sum(b[2] for b in batches) / len(batches), 3), "loss": round(loss, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--graphs", action="store_true", help="replay captured steps per batch shape (Model.graphed_train_step) instead of eager launches")
    ap.add_argument("--bucket", type=int, default=None, help="with --graphs: zero-pad AV batches to multiples of this many video frames")
    args = ap.parse_args()
    GRAPHS["on"], GRAPHS["bucket"] = args.graphs, args.bucket
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import avec_amd
    import nnet
    avec_amd.set_compute_dtype("bf16")
    avec_amd.manual_seed(1234)
    prec = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    want = lambda n: args.only in (None, n)
    out = []

    def av_model():
        torch.manual_seed(0)
        m = nnet.AudioVisualEfficientConformerInterCTC(vocab_size=256, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2])
        m.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
        return m.to(dev).train()

    if want("av15s") or want("lrs2_main") or want("lrs2_main_bucketed") or want("lrs2_pre"):
        model = av_model()
        if want("av15s"):
            out.append(run("av15s", model, [av_batch(8, torch.full((8,), 15.0), g, dev)], args.steps, args.warmup, prec))
            print(json.dumps(out[-1]), flush=True)
        if want("lrs2_main"):
            bs = [av_batch(32, durations(32, 2.0, 0.6, 0.8, 6.2, g), g, dev) for _ in range(4)]
            out.append(run("lrs2_main", model, bs, args.steps, args.warmup + 2, prec))
            print(json.dumps(out[-1]), flush=True)
        if want("lrs2_main_bucketed"):
            # the same length distribution through the length-bucketed batch sampler (avec_amd/nnet/samplers.py): 512 clips -> 16 batches of 32 neighbours in length
            from avec_amd.nnet.samplers import LengthBucketBatchSampler
            dur = durations(512, 2.0, 0.6, 0.8, 6.2, g)
            bsamp = LengthBucketBatchSampler(dur.tolist(), 32, shuffle=True, seed=0)
            bs = [av_batch(32, dur[torch.tensor(idx)], g, dev) for idx in list(bsamp)]
            out.append(run("lrs2_main_bucketed", model, bs, max(args.steps, len(bs)), args.warmup + 2, prec))
            print(json.dumps(out[-1]), flush=True)
        if want("lrs2_pre"):
            bs = [av_batch(16, durations(16, 6.0, 0.6, 0.8, 16.0, g), g, dev) for _ in range(4)]
            out.append(run("lrs2_pre", model, bs, args.steps, args.warmup + 2, prec))
            print(json.dumps(out[-1]), flush=True)
        del model
        torch.cuda.empty_cache()
    if want("ao"):
        torch.manual_seed(0)
        m = nnet.AudioEfficientConformerInterCTC(vocab_size=256, att_type="patch", interctc_blocks=[8, 11])
        m.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
        m = m.to(dev).train()
        B = 32
        audio, alen = 0.1 * torch.randn(B, 63840, generator=g), torch.full((B,), 63840)
        labels, llen = torch.randint(1, 256, (B, 20), generator=g), torch.full((B,), 20)
        out.append(run("ao", m, [([audio.to(dev), alen.to(dev)], (labels.to(dev), llen.to(dev)), 1.0, B)], args.steps, args.warmup, prec))
        print(json.dumps(out[-1]), flush=True)
        del m
    if want("vo"):
        torch.manual_seed(0)
        m = nnet.VisualEfficientConformerInterCTC(vocab_size=256, interctc_blocks=[3, 6])
        m.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
        m = m.to(dev).train()
        B = 32
        video, vlen = torch.randn(B, 100, 88, 88, 1, generator=g), torch.full((B,), 100)
        labels, llen = torch.randint(1, 256, (B, 20), generator=g), torch.full((B,), 20)
        out.append(run("vo", m, [([video.to(dev), vlen.to(dev)], (labels.to(dev), llen.to(dev)), 1.0, B)], args.steps, args.warmup, prec))
        print(json.dumps(out[-1]), flush=True)
        del m
    if want("lrw"):
        torch.manual_seed(0)
        m = nnet.VisualEfficientConformerCE(vocab_size=500)
        m.compile()
        m = m.to(dev).train()
        B = 64
        video = torch.randn(B, 1, 29, 88, 88, generator=g)
        labels = torch.randint(0, 500, (B,), generator=g)
        out.append(run("lrw", m, [(video.to(dev), labels.to(dev), 1.0, B)], args.steps, args.warmup, prec))
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
