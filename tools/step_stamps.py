"""Where does the graph-replayed training step spend its time, per stream, WITHOUT a profiler?  AVEC_STAMPS=1 makes the model launch one-wave kernels that write the wall
clock at named points of the forward / backward passes (ops.stamp / ops.mark); they are captured with the step and replay with it.      python tools/step_stamps.py"""
import os
import sys

os.environ["AVEC_STAMPS"] = "1"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import avec_amd
    import nnet
    from avec_amd import ops
    dev = torch.device("cuda:0")
    avec_amd.set_compute_dtype("bf16")
    avec_amd.manual_seed(1234)
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC(vocab_size=256, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2])
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    model = model.to(dev).train()
    inputs, targets = bench.synthetic_batch(32, dev, seed=0)
    step = model.make_graphed_train_step(inputs, targets, precision=torch.bfloat16, warmup=2)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    n = 10
    i0 = ops.STAMPS["names"].index("step_start:f")
    acc = 0
    for _ in range(n):
        step()
        torch.cuda.synchronize()
        t = ops.STAMPS["buf"][:len(ops.STAMPS["names"])].cpu().double()
        acc = acc + (t - t[i0])
    rel = acc / n / 100.0          # 100 MHz ticks -> us
    order = sorted(range(len(rel)), key=lambda i: rel[i])
    print("mark (forward :f / backward :b), microseconds after the first kernel of the visual front-end, mean of %d replays" % n)
    for i in order:
        print("%10.1f  %s" % (rel[i], ops.STAMPS["names"][i]))


if __name__ == "__main__":
    main()
