"""Dataset entry points named by the configs (nnet/datasets.py).  The licensed LRS2/LRS3 corpora are not available offline; when
their file lists are absent these classes yield synthetic LRS2-shaped samples (SURVEY 8d "secondary" distribution) so that
`main.py -c configs/LRS23/AV/EffConfInterCTC.py` runs end to end."""
import math
import os

import torch


class LRS(torch.utils.data.Dataset):
    def __init__(self, batch_size=None, collate_fn=None, version="LRS2", mode="test", root="datasets", shuffle=True, video_max_length=None,
                 audio_max_length=None, video_transform=None, audio_transform=None, align=True, num_synthetic=64, seed=0, **kwargs):
        self.batch_size, self.collate_fn, self.version, self.mode, self.shuffle = batch_size, collate_fn, version, mode, shuffle
        self.video_transform, self.audio_transform, self.align = video_transform, audio_transform, align
        self.synthetic = not os.path.exists(os.path.join(root, version))
        g = torch.Generator().manual_seed(seed)
        cap = (video_max_length or 400) / 25.0
        dur = torch.exp(math.log(2.0) + 0.6 * torch.randn(num_synthetic, generator=g)).clamp(0.8, min(6.2, cap))
        self.durations = dur.tolist()
        self.seed = seed

    def __len__(self):
        return len(self.durations)

    def __getitem__(self, n):
        g = torch.Generator().manual_seed(self.seed * 100003 + n)
        ta = int(16000 * self.durations[n])
        tv = ta // 640 + 1
        if self.video_transform is not None:
            # as the reference pipeline hands it over (nnet/datasets.py:187-196,348-352): normalised grayscale (1, T, 96, 96) -> config transform (crop 88x88, flip,
            # time masks) -> (T, 88, 88, 1).  The clip has the frames of its video track; align_video_to_audio pads to Ta // 640 + 1 afterwards
            video = self.video_transform(torch.randn(1, tv, 96, 96, generator=g)).permute(1, 2, 3, 0).contiguous()
        else:
            video = torch.randn(tv, 88, 88, 1, generator=g)
        audio = 0.1 * torch.randn(ta, generator=g)
        L = max(1, math.ceil(2.4 * self.durations[n]))
        label = torch.randint(1, 256, (L,), generator=g)
        return video, audio, label, torch.tensor(tv), torch.tensor(ta), torch.tensor(L)


class MultiDataset(torch.utils.data.ConcatDataset):
    def __init__(self, batch_size, collate_fn, datasets, shuffle=True):
        super().__init__(datasets)
        self.batch_size, self.collate_fn, self.shuffle = batch_size, collate_fn, shuffle
