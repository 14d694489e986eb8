"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's per-sample video input chain + collate, used to check
avec_amd.input_pipeline (the device pipeline).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows, in order:
  nnet/datasets.py:187-196   video_preprocessing = ConvertImageDtype(float32) -> Grayscale -> NormalizeVideo(mean, std) -> video_transform
  nnet/datasets.py:348-356   permute (T,H,W,C)->(C,T,H,W), preprocessing, align_video_to_audio
  configs/LRS23/AV/EffConfInterCTC.py:82-89   training: RandomCrop(88,88), RandomHorizontalFlip, TimeMaskSecond(0.4, 1.0, 25 fps, mean_frame); evaluation: CenterCrop
  nnet/transforms.py:40-52   NormalizeVideo            (pinned: tests/golden/video_input_ref.npz comes from the reference's own class)
  nnet/transforms.py:108-126 TimeMaskSecond            (loop count and fill value from the reference; the interval draw is torchaudio's)
  nnet/transforms.py:169-180 align_video_to_audio      (pinned by the same fixture)
  nnet/collate_fn.py:143-146 zero padding to the batch maximum

PARITY UNPINNED at the torchvision / torchaudio boundary (neither is installed here, the reference does not vendor or test them): ConvertImageDtype
(uint8 -> x / 255), Grayscale (0.2989 r + 0.587 g + 0.114 b), RandomCrop.get_params (two torch.randint draws, none when the clip already has the crop size),
RandomHorizontalFlip (torch.rand(1) < p), CenterCrop (int(round((H - h) / 2))) and torchaudio.functional.mask_along_axis (value = rand * mask_param,
min = rand * (T - value), [long(min), long(min) + long(value))) are restated from those libraries' published algorithms."""
import torch


def normalize_video(x, mean, std):
    m = torch.tensor(mean, dtype=torch.float32).reshape(len(mean), 1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).reshape(len(std), 1, 1, 1)
    return (x - m) / s


def align_video_to_audio(video, audio_len):
    tv, h, w, c = video.shape
    padding = audio_len // 640 + 1 - tv
    left, right = padding // 2, padding // 2 + padding % 2
    return torch.cat([video.new_zeros(left, h, w, c), video, video.new_zeros(right, h, w, c)], dim=0)


def mask_along_time(x, mask_param, mask_value):
    """x: (..., 1, T); torchaudio.functional.mask_along_axis(axis=2) on the (H*W, 1, T) view"""
    T = x.shape[-1]
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (T - value)
    start, end = int(min_value.long()), int(min_value.long() + value.long())
    keep = torch.ones(T, dtype=torch.bool)
    keep[start:end] = False
    return torch.where(keep, x, torch.as_tensor(mask_value, dtype=x.dtype)), (start, end)


def video_sample(video_u8, audio_len, train, crop=(88, 88), flip_p=0.5, T_second=0.4, num_mask_second=1.0, fps=25.0, mean_frame=True,
                 img_mean=(0.5,), img_std=(0.5,), align=True):
    """one LRS.__getitem__ video: uint8 (T,H,W,C) -> fp32 (T',h,w,1); consumes the global torch RNG exactly as the reference chain does"""
    x = video_u8.permute(3, 0, 1, 2).to(torch.float32) / 255          # (C,T,H,W)
    x = x.permute(1, 0, 2, 3)                                         # (T,C,H,W)
    if x.shape[1] == 3:
        r, g, b = x.unbind(dim=1)
        x = (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(1)
    x = normalize_video(x.permute(1, 0, 2, 3), img_mean, img_std)     # (1,T,H,W)
    H, W = x.shape[-2:]
    th, tw = crop
    masks = []
    if train:
        i = j = 0
        if not (H == th and W == tw):
            i = int(torch.randint(0, H - th + 1, size=(1,)).item())
            j = int(torch.randint(0, W - tw + 1, size=(1,)).item())
        x = x[..., i:i + th, j:j + tw]
        if torch.rand(1) < flip_p:
            x = x.flip(-1)
        x = x.permute(2, 3, 0, 1)                                     # (h,w,1,T)
        for _ in range(int(x.shape[-1] / fps * num_mask_second)):
            x, se = mask_along_time(x, int(T_second * fps), x.mean() if mean_frame else 0.0)
            masks.append(se)
        x = x.permute(2, 3, 0, 1)
    else:
        i, j = int(round((H - th) / 2.0)), int(round((W - tw) / 2.0))
        x = x[..., i:i + th, j:j + tw]
    x = x.permute(1, 2, 3, 0)                                         # (T,h,w,1)
    if align:
        x = align_video_to_audio(x, audio_len)
    return x, masks


def video_batch(clips, audio_lens, train, **kw):
    """the list of samples a dataloader worker would hand to CollateFn, zero-padded on the time axis; returns (B,Tmax,h,w,1), lengths"""
    outs = [video_sample(c, int(a), train, **kw)[0] for c, a in zip(clips, audio_lens)]
    lens = torch.tensor([o.shape[0] for o in outs], dtype=torch.long)
    T = int(lens.max())
    batch = torch.zeros(len(outs), T, *outs[0].shape[1:])
    for b, o in enumerate(outs):
        batch[b, :o.shape[0]] = o
    return batch, lens
