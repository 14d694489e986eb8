"""Video input pipeline on the device: decoded uint8 mouth clips -> the (B, T, 88, 88, 1) fp32 batch the visual front-end reads.

The reference does this per sample on dataloader workers (nnet/datasets.py:187-196,348-356; AV cfg:82-89; nnet/transforms.py:40-52,108-126,169-180) and zero-pads
in CollateFn (nnet/collate_fn.py:143-146): ~0.8 M pixels per utterance through five torch ops and a Python loop.  Here only the random DECISIONS stay on the host
(crop origin, flip, time-mask intervals: a handful of scalars per clip, drawn from the torch RNG in the reference's call order so that a seeded run picks the
same augmentations); the clips cross PCIe once as uint8 (a quarter of the fp32 bytes) from a pinned staging buffer, and three HBM-bound launches
(csrc/video_input.hip, `avec_video_input`) produce the whole normalised, cropped, flipped, masked, aligned and padded batch."""
import torch

from . import runtime as rt
from .lib import lib


class VideoInputPipeline:
    def __init__(self, crop_size=(88, 88), training=True, flip_p=0.5, T_second=0.4, num_mask_second=1.0, fps=25.0, mean_frame=True, img_mean=0.5, img_std=0.5,
                 align=True, device="cuda"):
        self.crop, self.training, self.flip_p = tuple(crop_size), training, flip_p
        self.T_mask, self.num_mask_second, self.fps, self.mean_frame = int(T_second * fps), num_mask_second, fps, mean_frame
        self.mean, self.std, self.align = float(img_mean), float(img_std), align
        self.device = torch.device(device)
        self._stage = None          # pinned uint8 staging buffer, grown on demand
        self._lut = {}

    # ---- host side: the random decisions, in the reference's RNG call order for one sample after the other -------------------------------------------
    def draw(self, shapes, audio_lens=None):
        """shapes: [(T, H, W, C)] per clip.  Returns geom int32 [B][8] = (tv, H, W, crop_y, crop_x, flip, pad_left, n_masks), masks int32 [B][M][2], out lengths."""
        th, tw = self.crop
        B = len(shapes)
        geom = torch.zeros(B, 8, dtype=torch.int32)
        mlist, lens = [], []
        for b, (T, H, W, C) in enumerate(shapes):
            assert H >= th and W >= tw, "clip smaller than the crop"
            ms = []
            if self.training:
                cy = cx = 0
                if not (H == th and W == tw):                              # torchvision RandomCrop.get_params
                    cy = int(torch.randint(0, H - th + 1, size=(1,)).item())
                    cx = int(torch.randint(0, W - tw + 1, size=(1,)).item())
                flip = int(bool(torch.rand(1) < self.flip_p))              # RandomHorizontalFlip
                for _ in range(int(T / self.fps * self.num_mask_second)):  # TimeMaskSecond -> torchaudio mask_along_axis
                    value = torch.rand(1) * self.T_mask
                    lo = torch.rand(1) * (T - value)
                    ms.append((int(lo.long()), int(lo.long() + value.long())))
            else:                                                          # CenterCrop
                cy, cx, flip = int(round((H - th) / 2.0)), int(round((W - tw) / 2.0)), 0
            pad_left, total = 0, T
            if self.align:                                                 # align_video_to_audio: zero frames split left / right
                padding = int(audio_lens[b]) // 640 + 1 - T
                assert padding >= 0, "clip longer than its audio track allows (Tv > Ta // 640 + 1)"
                pad_left, total = padding // 2, T + padding
            geom[b] = torch.tensor([T, H, W, cy, cx, flip, pad_left, len(ms)], dtype=torch.int32)
            mlist.append(ms)
            lens.append(total)
        M = max([len(m) for m in mlist] + [0])
        masks = torch.zeros(B, max(M, 1), 2, dtype=torch.int32)
        for b, ms in enumerate(mlist):
            for k, se in enumerate(ms):
                masks[b, k] = torch.tensor(se, dtype=torch.int32)
        return geom, masks, M, torch.tensor(lens, dtype=torch.long)

    def lut(self, channels):
        """[channels][256] fp32 = w_c * (u / 255): ConvertImageDtype and the Grayscale products, by the same fp32 torch operations the reference chain applies"""
        if channels not in self._lut:
            x = torch.arange(256, dtype=torch.uint8).to(torch.float32) / 255
            rows = [x] if channels == 1 else [0.2989 * x, 0.587 * x, 0.114 * x]
            self._lut[channels] = torch.stack(rows).contiguous().to(self.device)
        return self._lut[channels]

    # ---- staging: one pinned buffer, one asynchronous copy --------------------------------------------------------------------------------------------
    def stage(self, clips):
        """clips: list of uint8 (T,H,W,C) tensors (host or device).  Returns (device uint8 flat buffer, byte offsets int64 [B])."""
        sizes = [c.numel() for c in clips]
        offs = [0]
        for n in sizes[:-1]:
            offs.append(offs[-1] + (n + 15) // 16 * 16)
        total = offs[-1] + sizes[-1]
        if all(c.is_cuda for c in clips):
            flat = torch.empty(total, dtype=torch.uint8, device=self.device)
            for c, o, n in zip(clips, offs, sizes):
                flat[o:o + n].copy_(c.reshape(-1))
        else:
            if self._stage is None or self._stage.numel() < total:
                self._stage = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8).pin_memory()
            for c, o, n in zip(clips, offs, sizes):
                assert c.dtype == torch.uint8
                self._stage[o:o + n].copy_(c.reshape(-1))
            flat = self._stage[:total].to(self.device, non_blocking=True)
        return flat, torch.tensor(offs, dtype=torch.long)

    def __call__(self, clips, audio_lens=None, params=None):
        """-> video fp32 (B, Tmax, h, w, 1) on the device, video_len int64 (B,) (host)"""
        assert self.device.type == "cuda", "the video input pipeline runs on the GPU (no host fallback)"
        shapes = [tuple(c.shape) for c in clips]
        C = shapes[0][3]
        assert all(s[3] == C for s in shapes) and C in (1, 3)
        geom, masks, M, lens = params if params is not None else self.draw(shapes, audio_lens)
        flat, offs = self.stage(clips)
        B, Tout = len(clips), int(lens.max())
        th, tw = self.crop
        geom_d, offs_d = geom.to(self.device, non_blocking=True), offs.to(self.device, non_blocking=True)
        masks_d = masks.to(self.device, non_blocking=True) if M else None
        out = torch.empty(B, Tout, th, tw, 1, dtype=torch.float32, device=self.device)
        ws = torch.empty(2 * B * Tout, dtype=torch.float32, device=self.device) if M else None
        lib.video_input(flat.data_ptr(), offs_d.data_ptr(), geom_d.data_ptr(), None if masks_d is None else masks_d.data_ptr(), M, C, self.lut(C).data_ptr(), self.mean, self.std,
                        int(self.mean_frame), out.data_ptr(), None if ws is None else ws.data_ptr(), B, Tout, th, tw, rt.stream())
        return out, lens
