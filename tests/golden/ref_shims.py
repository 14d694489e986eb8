"""Import the reference implementation (`/root/reference/nnet`) in THIS container only.

The reference hard-imports packages that are not installed here (tensorboard, torchaudio,
torchvision, jiwer, skimage, gdown).  This module registers minimal stand-ins in `sys.modules`
so that `import nnet` (the reference's) succeeds, and returns the package.

Only `tests/golden/make_golden.py` (fixture generator) and `tests/golden/check_oracle_fullsize.py`
use this; nothing here travels to the GPU box as a dependency of any test (`/root/reference` does
not exist there).  The torchaudio stand-ins are a restatement of torchaudio's *documented*
defaults (torchaudio is un-vendored and unpinned in the reference: requirements.txt:2):
  Spectrogram(n_fft, win_length, hop_length): torch.stft(center=True, pad_mode="reflect",
      window=hann_window(win_length) (periodic), onesided) -> |.|^2
  MelScale(n_mels, sample_rate, f_min, f_max, n_stft, norm=None, mel_scale="htk").
"""
import math
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def melscale_fbanks_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


class _Spectrogram(nn.Module):
    def __init__(self, n_fft=400, win_length=None, hop_length=None):
        super().__init__()
        self.n_fft = n_fft
        self.win_length = win_length or n_fft
        self.hop_length = hop_length or self.win_length // 2
        self.register_buffer("window", torch.hann_window(self.win_length), persistent=False)

    def forward(self, x):
        spec = torch.stft(x, self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                          window=self.window, center=True, pad_mode="reflect", normalized=False,
                          onesided=True, return_complex=True)
        return spec.abs().pow(2.0)


class _MelScale(nn.Module):
    def __init__(self, n_mels=128, sample_rate=16000, f_min=0.0, f_max=None, n_stft=201):
        super().__init__()
        f_max = f_max if f_max is not None else float(sample_rate // 2)
        self.register_buffer("fb", melscale_fbanks_htk(n_stft, f_min, f_max, n_mels, sample_rate),
                             persistent=False)

    def forward(self, spec):
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


class _Identity(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x, *a, **k):
        return x


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns the reference `nnet` package (imported from /root/reference)."""
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    class _SummaryWriter:
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_text(self, *a, **k): pass
        def flush(self): pass

    tb = _mod("tensorboard")
    tut = _mod("torch.utils.tensorboard", SummaryWriter=_SummaryWriter)
    torch.utils.tensorboard = tut

    ta_t = _mod("torchaudio.transforms", Spectrogram=_Spectrogram, MelScale=_MelScale,
                RNNTLoss=nn.Module, FrequencyMasking=_Identity, TimeMasking=_Identity)
    ta_f = _mod("torchaudio.functional", mask_along_axis=lambda x, *a, **k: x)
    _mod("torchaudio", transforms=ta_t, functional=ta_f, load=None)

    tv_t = _mod("torchvision.transforms", RandomCrop=_Identity, RandomHorizontalFlip=_Identity,
                CenterCrop=_Identity)
    tv_du = _mod("torchvision.datasets.utils", extract_archive=None)
    tv_d = _mod("torchvision.datasets", utils=tv_du)
    tv_io = _mod("torchvision.io")
    _mod("torchvision", transforms=tv_t, datasets=tv_d, io=tv_io)

    _mod("jiwer", wer=lambda *a, **k: 0.0)
    sk_t = _mod("skimage.transform")
    _mod("skimage", transform=sk_t)
    _mod("gdown")
    _mod("av")

    assert "nnet" not in sys.modules or getattr(sys.modules["nnet"], "__file__", "").startswith(
        REFERENCE_ROOT), "another package named nnet is already imported"
    import nnet  # noqa: E402  (the reference's)
    assert nnet.__file__.startswith(REFERENCE_ROOT)
    return nnet
