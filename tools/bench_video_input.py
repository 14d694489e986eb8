"""Throughput of the device video input pipeline (avec_video_input) on one MI355X: B clips of T frames, 96x96 uint8 -> (B,T',88,88,1) fp32.

    python tools/bench_video_input.py [--batch 32 --frames 100 --channels 3]

Reports (a) the three launches alone with the clips resident in HBM (HIP events), against the HBM roofline with the algorithmic bytes
(crop-window read + fp32 write per frame), (b) the same including packing into the pinned buffer + the H2D copy, (c) the reference's per-sample chain
(oracle/video_input.py, CPU, torch threads as reported) on a few clips, as utterances/s."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from avec_amd.input_pipeline import VideoInputPipeline
    from oracle import video_input as VO
    torch.manual_seed(0)
    clips = [torch.randint(0, 256, (a.frames, 96, 96, a.channels), dtype=torch.uint8) for _ in range(a.batch)]
    alens = [a.frames * 640 - 1] * a.batch
    pipe = VideoInputPipeline(training=True, device="cuda")
    dev_clips = [c.cuda() for c in clips]
    params = pipe.draw([tuple(c.shape) for c in clips], alens)
    for _ in range(3):
        pipe(dev_clips, alens, params=params)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # (a) kernels only: stage once, then call the C entry point directly
    from avec_amd.lib import lib
    from avec_amd import runtime as rt
    flat, offs = pipe.stage(dev_clips)
    geom, masks, M, lens = params
    gd, od, md = geom.cuda(), offs.cuda(), masks.cuda()
    B, Tout = a.batch, int(lens.max())
    out = torch.empty(B, Tout, 88, 88, 1, device="cuda")
    ws = torch.empty(2 * B * Tout, device="cuda")
    st = torch.cuda.current_stream()
    e0.record(st)
    for _ in range(a.iters):
        lib.video_input(flat.data_ptr(), od.data_ptr(), gd.data_ptr(), md.data_ptr(), M, a.channels, pipe.lut(a.channels).data_ptr(), 0.5, 0.5, 1, out.data_ptr(), ws.data_ptr(),
                        B, Tout, 88, 88, rt.stream())
    e1.record(st)
    torch.cuda.synchronize()
    ms_k = e0.elapsed_time(e1) / a.iters
    bytes_alg = B * a.frames * (88 * 88 * a.channels + 88 * 88 * 4)
    # (b) from pageable host clips: pack into the pinned buffer + H2D + kernels
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        pipe(clips, alens, params=params)
    torch.cuda.synchronize()
    ms_h = 1e3 * (time.perf_counter() - t0) / a.iters
    # (c) the per-sample CPU chain
    n = min(4, a.batch)
    t0 = time.perf_counter()
    for c, l in zip(clips[:n], alens[:n]):
        VO.video_sample(c, l, True)
    cpu_s = (time.perf_counter() - t0) / n
    print(json.dumps({"workload": "video input, %d clips x %d frames, 96x96x%d uint8 -> 88x88 fp32, train augmentations" % (a.batch, a.frames, a.channels),
                      "kernels_ms": round(ms_k, 4), "kernels_utt_per_s": round(B / ms_k * 1e3), "algorithmic_GBps": round(bytes_alg / ms_k / 1e6, 1),
                      "hbm_frac_of_8TBps": round(bytes_alg / ms_k / 1e6 / 8000, 3),
                      "host_to_batch_ms": round(ms_h, 3), "host_to_batch_utt_per_s": round(B / ms_h * 1e3),
                      "cpu_chain_utt_per_s": round(1 / cpu_s, 1), "torch_threads": torch.get_num_threads()}))


if __name__ == "__main__":
    main()
