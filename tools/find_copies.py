"""Which ops.py / nnet lines issue the device-to-device copies, fills and other ATen kernels of one training step?
One eager step under torch.profiler with Python stacks; prints the ATen ops (copy_/fill_/zero_/add/...) aggregated by the innermost frame
that lies inside this repository.      python tools/find_copies.py [--batch 32]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    import avec_amd
    import nnet
    dev = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    avec_amd.manual_seed(1234)
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC(vocab_size=256, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2])
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    model = model.to(dev).train()
    inputs, targets = bench.synthetic_batch(args.batch, dev, seed=0)
    for _ in range(3):
        model.train_step(inputs, targets, precision=torch.bfloat16)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        model.train_step(inputs, targets, precision=torch.bfloat16)
        torch.cuda.synchronize()
    agg = collections.Counter()
    shapes = {}
    for ev in prof.events():
        if not ev.name.startswith("aten::"):
            continue
        if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
            continue                       # only outermost ATen calls
        frame = "?"
        for fr in (ev.stack or []):
            if ROOT in fr and "tools/find_copies" not in fr:
                frame = fr.replace(ROOT + "/", "")
                break
        key = (ev.name, frame)
        agg[key] += 1
        shapes.setdefault(key, str(ev.input_shapes)[:90])
    print("outermost ATen calls of one training step, by repository frame (count, op, frame, example shapes):")
    for (name, frame), n in agg.most_common(80):
        print("%5d  %-28s %-70s %s" % (n, name, frame[:70], shapes[(name, frame)]))
    kern = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            kern[ev.name[:80]] += 1
    print("\ndevice activities (count, name):")
    for k, n in kern.most_common(25):
        print("%5d  %s" % (n, k))


if __name__ == "__main__":
    main()
