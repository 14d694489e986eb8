"""Stress of the audio front (mel spectrogram + Conv2d/BatchNorm/Swish stem, batch 1) as tools/ddp_equiv.py runs it with eight ranks on one GPU: N processes share the GPU,
each repeats the front on a side stream (the visual stem on the main stream beside it) and compares every result with its first one.  Prints one line per process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    tag = sys.argv[2] if len(sys.argv) > 2 else "0"
    dev = torch.device("cuda", 0)
    import avec_amd, nnet
    from avec_amd import ops, runtime as rt
    avec_amd.set_compute_dtype(os.environ.get("STRESS_DTYPE", "f32"))
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC()
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    model = model.to(dev).train()
    enc = model.encoder.audio_encoder
    venc = model.encoder.video_encoder
    g = torch.Generator().manual_seed(5)
    video, audio = torch.randn(1, 20, 88, 88, 1, generator=g), 0.1 * torch.randn(1, 12160, generator=g)
    audio, alen = audio.to(dev), torch.tensor([12160], device=dev)
    video = video.to(dev).permute(0, 4, 1, 2, 3).contiguous()
    side = torch.cuda.Stream()
    stem = enc.subsampling_module.layers[0]
    first, bad = None, []
    for it in range(iters):
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        with torch.no_grad():
            if os.environ.get("STRESS_VIDEO", "1") == "1":
                venc.forward_front(video)                              # main stream: the visual front-end beside the audio branch
        with torch.cuda.stream(side):
            mel, _ = enc.audio_preprocessing(audio, alen)
            a = ops.AudioStemFn.apply(mel, stem[0].weight, stem[0], stem[1], True)
            st = a.grad_fn.saved[2]
            cur = (mel.clone(), st.stats[:2 * st.C].clone(), a.detach().float().clone())
        main_s.wait_stream(side)
        torch.cuda.synchronize()
        if first is None:
            first = cur
            continue
        for name, x, y in zip(("mel", "stats", "out"), first, cur):
            d = (x - y).abs().max().item()
            if not d <= 1e-4 * x.abs().max().item():
                bad.append((it, name, d, x.abs().max().item(), int(((x - y).abs() > 1e-4 * x.abs().max()).sum())))
    print("proc %s: %d iterations, %d deviations %s | mel %.9g stats %.9g out %.9g" % (tag, iters, len(bad), bad[:6], first[0].double().sum().item(), first[1].double().sum().item(), first[2].double().sum().item()), flush=True)


if __name__ == "__main__":
    main()
