"""Minimal `torchvision.transforms` subset, restated from torchvision's published behaviour (the package is neither installed nor vendored by the
reference): the random draws follow its call order -- RandomCrop.get_params: i = randint(0, h - th + 1), then j = randint(0, w - tw + 1);
RandomHorizontalFlip: one torch.rand(1) < p -- so a seeded run picks the same augmentations as avec_amd.input_pipeline and the oracle."""
import types

import torch
import torch.nn as nn


class RandomCrop(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def forward(self, img):
        h, w = img.shape[-2:]
        th, tw = self.size
        assert h >= th and w >= tw, "crop %s larger than the image (%d, %d)" % (self.size, h, w)
        if h == th and w == tw:
            return img
        i = int(torch.randint(0, h - th + 1, size=(1,)).item())
        j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        return img[..., i:i + th, j:j + tw]


class CenterCrop(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def forward(self, img):
        h, w = img.shape[-2:]
        th, tw = self.size
        i, j = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return img[..., i:i + th, j:j + tw]


class RandomHorizontalFlip(nn.Module):
    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, img):
        return img.flip(-1) if torch.rand(1) < self.p else img


class ConvertImageDtype(nn.Module):
    def __init__(self, dtype):
        super().__init__()
        self.dtype = dtype

    def forward(self, img):
        if img.dtype == torch.uint8 and self.dtype.is_floating_point:
            return img.to(self.dtype) / 255.0
        return img.to(self.dtype)


class Grayscale(nn.Module):
    def __init__(self, num_output_channels=1):
        super().__init__()
        assert num_output_channels == 1

    def forward(self, img):                                   # (..., C, H, W)
        if img.shape[-3] == 1:
            return img
        r, g, b = img.unbind(dim=-3)
        return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(-3)


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


def build():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    for cls in (RandomCrop, CenterCrop, RandomHorizontalFlip, ConvertImageDtype, Grayscale, Compose):
        setattr(tr, cls.__name__, cls)
    tv.transforms = tr
    tv.__version__ = "0.0+avec_amd.fallback"
    return tv
