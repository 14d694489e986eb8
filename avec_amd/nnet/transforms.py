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


def align_video_to_audio(video, audio, video_frames_per_audio=640):
    """nnet/transforms.py:169-180: Tv = Ta // 640 + 1 (crop or zero-pad the frames)."""
    tv = audio.shape[-1] // video_frames_per_audio + 1
    if video.shape[0] >= tv:
        return video[:tv]
    return torch.cat([video, video.new_zeros(tv - video.shape[0], *video.shape[1:])], dim=0)


class TimeMaskSecond(nn.Module):
    """host-side augmentation named by the AV config (nnet/transforms.py:108-126): per second of video, mask up to T_second seconds."""

    def __init__(self, T_second, num_mask_second, fps, mean_frame=False):
        super().__init__()
        self.T = int(T_second * fps)
        self.num_mask_second, self.fps, self.mean_frame = num_mask_second, fps, mean_frame

    def forward(self, x):
        n = int(self.num_mask_second * x.shape[-1] / self.fps)
        for _ in range(n):
            w = int(torch.randint(0, self.T + 1, (1,)))
            if w == 0 or x.shape[-1] - w <= 0:
                continue
            s = int(torch.randint(0, x.shape[-1] - w, (1,)))
            x[..., s:s + w] = x.mean() if self.mean_frame else 0.0
        return x
