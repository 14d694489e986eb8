"""Reshape helpers used inside the visual encoder + host-side data transforms named by the configs (nnet/transforms.py)."""
import torch
import torch.nn as nn


def video_to_images(videos):
    return videos.transpose(1, 2).flatten(start_dim=0, end_dim=1)


def images_to_videos(images, video_frames):
    assert images.size(0) % video_frames == 0
    return images.reshape(images.size(0) // video_frames, video_frames, *images.shape[1:]).transpose(1, 2)


class VideoToImages(nn.Module):
    def forward(self, videos):
        return video_to_images(videos)


class ImagesToVideos(nn.Module):
    def __init__(self, video_frames=None):
        super().__init__()
        self.video_frames = video_frames

    def forward(self, images, video_frames=None):
        return images_to_videos(images, self.video_frames if video_frames is None else video_frames)


class NormalizeVideo(nn.Module):
    """nnet/transforms.py:40-52 (host-side; the batched device version is avec_amd.input_pipeline)"""

    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean, dtype=torch.float32).reshape(len(mean), 1, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor(std, dtype=torch.float32).reshape(len(std), 1, 1, 1), persistent=False)

    def forward(self, x):
        return (x - self.mean) / self.std


def align_video_to_audio(video, audio, video_frames_per_audio=640):
    """nnet/transforms.py:169-180: the clip (Tv,H,W,C) gets Ta // 640 + 1 - Tv zero frames, half in front (floor), the rest behind."""
    tv = video.shape[0]
    padding = audio.shape[0] // video_frames_per_audio + 1 - tv
    left, right = padding // 2, padding // 2 + padding % 2
    return torch.cat([video.new_zeros(left, *video.shape[1:]), video, video.new_zeros(right, *video.shape[1:])], dim=0)


class TimeMaskSecond(nn.Module):
    """host-side augmentation named by the AV config (nnet/transforms.py:108-126): int(T / fps * num_mask_second) masks on the last axis, each drawn as
    torchaudio.functional.mask_along_axis does (width = rand * T_mask, start = rand * (T - width), both truncated) and filled with the mean of the clip as it is
    at that moment (mean_frame) or 0.  The batched device version is avec_amd.input_pipeline."""

    def __init__(self, T_second, num_mask_second, fps, mean_frame=False):
        super().__init__()
        self.T = int(T_second * fps)
        self.num_mask_second, self.fps, self.mean_frame = num_mask_second, fps, mean_frame

    def forward(self, x):
        T = x.shape[-1]
        for _ in range(int(T / self.fps * self.num_mask_second)):
            fill = x.mean() if self.mean_frame else 0.0
            width = torch.rand(1) * self.T
            lo = torch.rand(1) * (T - width)
            s, e = int(lo.long()), int(lo.long() + width.long())
            x = x.clone()
            x[..., s:e] = fill
        return x
