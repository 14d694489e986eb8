"""Audio front-end (nnet/preprocessing.py).  torchaudio is not a dependency: Spectrogram / MelScale are restated from torchaudio's
documented defaults (see oracle/avec_oracle.py) and executed as framing -> exact-fp32 MFMA DFT GEMM -> power/mel/log kernels."""
import math

import torch
import torch.nn as nn

from .. import ops
from .. import runtime as rt


class _Spectrogram(nn.Module):
    """holds the `window` buffer under the key real torchaudio checkpoints use (...audio_preprocessing.Spectrogram.window)"""

    def __init__(self, n_fft, win_length, hop_length):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        self.register_buffer("window", torch.hann_window(win_length), persistent=True)


class _MelScale(nn.Module):
    """holds `fb` (n_freqs, n_mels): HTK mel triangles, norm=None (...audio_preprocessing.MelScale.fb)"""

    def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
        super().__init__()
        all_freqs = torch.linspace(0, sample_rate // 2, n_stft)
        hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
        m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
        f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        fb = torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
        self.register_buffer("fb", fb, persistent=True)


class AudioPreprocessing(nn.Module):
    """(B, L) waveform -> (B, n_mels, L // hop + 1) log-mel, always fp32 (nnet/preprocessing.py:24-85)."""

    def __init__(self, sample_rate=16000, n_fft=512, win_length_ms=25, hop_length_ms=10, n_mels=80, normalize=False, mean=0, std=1):
        super().__init__()
        self.win_length = int(sample_rate * win_length_ms) // 1000
        self.hop_length = int(sample_rate * hop_length_ms) // 1000
        self.n_fft, self.n_mels = n_fft, n_mels
        self.Spectrogram = _Spectrogram(n_fft, self.win_length, self.hop_length)
        self.MelScale = _MelScale(n_mels, sample_rate, 0, 8000, n_fft // 2 + 1)
        self.normalize, self.mean, self.std = normalize, mean, std
        self._dft = None

    def _dft_matrix(self, device):
        """[2*nb][win]: rows k < nb = cos(2 pi k (n+left) / n_fft), rows nb+k = sin(...) over the non-zero window support."""
        if self._dft is None or self._dft.device != device:
            nb, left = self.n_fft // 2 + 1, (self.n_fft - self.win_length) // 2
            n = torch.arange(self.win_length, dtype=torch.float64) + left
            k = torch.arange(nb, dtype=torch.float64)
            ang = 2.0 * math.pi * k[:, None] * n[None, :] / self.n_fft
            self._dft = torch.cat([ang.cos(), ang.sin()], dim=0).float().to(device).contiguous()
        return self._dft

    def forward(self, x, lengths=None):
        dtype = x.dtype
        with torch.no_grad():
            mel = ops.mel_spectrogram(x.float(), self.Spectrogram.window, self._dft_matrix(x.device), self.MelScale.fb,
                                      self.n_fft, self.win_length, self.hop_length, self.n_mels)
        if lengths is not None:
            lengths = ops.len_affine(lengths, 0, self.hop_length, 1)
        if self.normalize:
            mel = (mel - self.mean) / self.std
        mel = mel.type(dtype) if dtype.is_floating_point else mel
        return (mel, lengths) if lengths is not None else mel


class SpecAugment(nn.Module):
    """nnet/preprocessing.py:87-130: mF batch-shared frequency masks (< F bins), mT per-sample time masks (< pS * length); train only.
    One kernel, counter-based RNG, no per-sample host loop / device syncs."""

    def __init__(self, mF, F, mT, pS):
        super().__init__()
        self.mF, self.F, self.mT, self.pS = mF, F, mT, pS
        self.rng_stream = rt.new_stream_id()

    def forward(self, samples, lengths):
        if self.training:
            samples = ops.spec_augment_(samples.float().contiguous(), lengths, self.mF, self.F, self.mT, self.pS, self.rng_stream)
        return samples
