"""One forward + backward (+ Adam) pass of the audio-visual model on the guard-page allocator (tools/guard/guard_alloc.cpp): any kernel that touches memory past the guarded
side of a tensor faults.  Test infrastructure.   GUARD_MODE=tail|head python tools/guard/guard_pass.py [--batch B] [--dtype bf16|f32] [--dist] [--steps N]"""
import argparse, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--no-guard", action="store_true")
    ap.add_argument("--frames", type=int, default=20, help="video frames per utterance (audio: 640 samples per frame minus a ragged tail)")
    ap.add_argument("--eval", action="store_true", help="after the training steps: one evaluation step (eval-mode BatchNorm, greedy CTC decoding of the outputs)")
    args = ap.parse_args()
    so = os.path.join(HERE, "libguard_alloc.so")
    if not args.no_guard:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "guard_alloc.cpp")):
            subprocess.run(["hipcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "guard_alloc.cpp")], check=True)
        alloc = torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free")
        torch.cuda.memory.change_current_allocator(alloc)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if args.dist:
        os.environ["AVEC_DIST_SINGLE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29589")
        torch.distributed.init_process_group(backend="gloo", rank=0, world_size=1)
    import avec_amd, nnet
    if os.environ.get("GUARD_TRACE"):
        # name every library call and wait for it, so that a fault is reported right after the entry point that caused it
        from avec_amd import lib as _libmod
        L = _libmod.lib
        orig_getattr = type(L).__getattr__
        def traced(self, name):
            fn = orig_getattr(self, name)
            def call(*a):
                sys.stderr.write("[guard] %s ... " % name); sys.stderr.flush()
                fn(*a)
                torch.cuda.synchronize()
                k = self.raw("avec_last_kernel")()
                sys.stderr.write("ok (%s)\n" % (k.decode() if isinstance(k, bytes) else k)); sys.stderr.flush()
            self.__dict__[name] = call
            return call
        type(L).__getattr__ = traced
        for k_ in [k_ for k_, v_ in L.__dict__.items() if callable(v_) and k_ != "_dll"]:
            del L.__dict__[k_]
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC()
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    model = model.to(dev).train()
    if args.dist:
        model.distribute_strategy(0)
    if args.dist and os.environ.get("GUARD_CHECK_EXCHANGE"):
        # one rank: every exchange must return exactly what went in
        from avec_amd import peer as _peer
        px = _peer.active()
        print("peer exchange active: %s" % (px is not None), flush=True)
        if px is not None:
            f1, nbad = px.all_reduce_sum_fused, [0]
            def checked(src, nrep, n_in, tail, key, dgamma=None, dbeta=None, C=0):
                torch.cuda.synchronize()
                loc = src[:nrep * n_in].view(nrep, n_in).sum(0)
                if tail is not None:
                    loc = torch.cat([loc, torch.full((1,), float(tail), device=loc.device)])
                out = f1(src, nrep, n_in, tail, key, dgamma, dbeta, C)
                torch.cuda.synchronize()
                bad = ~((out - loc).abs() <= 1e-6 * loc.abs() + 1e-30)
                if bool(bad.any()) and nbad[0] < 8:
                    nbad[0] += 1
                    i = int(bad.nonzero()[0])
                    print("exchange site %d (%s) n=%d: %d wrong, first [%d] got %r want %r; err flag %d epoch %d" % (px.sites[key], key[1] if isinstance(key, tuple) else key, loc.numel(), int(bad.sum()), i,
                          float(out[i]), float(loc[i]), int(px.err.item()), int(px.epochs[px.sites[key]].item())), flush=True)
                return out
            px.all_reduce_sum_fused = checked
    B = args.batch
    g = torch.Generator().manual_seed(5)
    T = args.frames
    L = 12160 if T == 20 else T * 640 - 160         # 20 frames <-> 12160 samples as everywhere in the tests; 100 frames <-> 63 840 samples as in bench.py
    video, audio = torch.randn(B, T, 88, 88, 1, generator=g), 0.1 * torch.randn(B, L, generator=g)
    vlen = torch.tensor(([T, max(T - 3, 1), T, max(T - 9, 1)] * B)[:B])
    alen = torch.tensor(([L, max(L - 2160, 400), L, max(L - 5160, 400)] * B)[:B])
    nl = max(2, min(4, T // 6))
    labels, llen = torch.randint(1, 256, (B, nl), generator=g), torch.tensor(([nl, nl - 1, nl, max(nl - 2, 1)] * B)[:B])
    inputs = [t.to(dev) for t in (video, vlen, audio, alen)]
    targets = (labels.to(dev), llen.to(dev))
    precision = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    for step in range(args.steps):
        losses, _, _ = model.train_step(inputs, targets, precision=precision)          # eager: forward, backward, (all-reduce,) Adam
        torch.cuda.synchronize()
        print("step %d loss %.6f" % (step, float(losses["loss"])), flush=True)
    if args.eval:
        model.eval()
        losses, metrics, truths, preds = model.eval_step(inputs, targets)
        torch.cuda.synchronize()
        print("eval loss %.6f, %d hypotheses" % (float(losses["loss"]), len(preds) if preds is not None else 0), flush=True)
    if not args.no_guard:
        import ctypes
        lib = ctypes.CDLL(so)
        lib.guard_stats.restype = ctypes.c_longlong
        print("GUARD PASS OK mode=%s allocations=%d mapped=%.1f MB granule=%d" % (os.environ.get("GUARD_MODE", "tail"), lib.guard_stats(0), lib.guard_stats(1) / 2 ** 20, lib.guard_stats(2)), flush=True)
    else:
        print("PASS OK (no guard)", flush=True)


if __name__ == "__main__":
    main()
