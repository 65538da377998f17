"""Waveform -> normalised log-mel spectrogram on the device, host side of `avsd_kaldi_fbank`.

Mirrors `waveform_to_melspectrogram` / `AudioMelspectrogramExtractor` of the reference (avgen/data/utils.py:26-110):
centre-crop the waveform to `clip_duration` seconds, ImageBind `waveform2melspec` (un-vendored submodule
facebookresearch/ImageBind, imagebind/data.py: `waveform -= waveform.mean()`, then
`torchaudio.compliance.kaldi.fbank(htk_compat=True, sample_frequency=sr, use_energy=False, window_type="hanning",
num_mel_bins, dither=0.0, frame_length=25, frame_shift=10)`, transpose to (mel, time), zero-pad / crop to
`target_length` frames) and `Normalize(mean, std)`.

Neither ImageBind nor torchaudio is in the reference tree or in this image: the filterbank below restates the published
Kaldi algorithm with torchaudio's defaults (povey-free hanning window, pre-emphasis 0.97, per-frame DC removal,
snip_edges, power spectrum, 20 Hz .. Nyquist mel triangles on the 1127 ln(1 + f / 700) scale, log floor at float32
eps) — parity unpinned (oracle/audio_ref.py says the same).  The whole-clip mean subtraction of waveform2melspec is
dropped: the per-frame DC removal that follows cancels any constant offset exactly.
"""
from __future__ import annotations

from typing import List, Union

import numpy as np
import torch

from . import ops


def _mel(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def kaldi_mel_banks(num_bins: int = 128, nfft: int = 512, sample_freq: float = 16000.0, low_freq: float = 20.0,
                    high_freq: float = 0.0) -> np.ndarray:
    """[num_bins][nfft/2 + 1] float32 triangular filters (torchaudio `get_mel_banks` + the zero Nyquist column)."""
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    n_fft_bins = nfft // 2
    bin_width = sample_freq / nfft
    mel_low, mel_high = _mel(low_freq), _mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    left = mel_low + np.arange(num_bins, dtype=np.float64)[:, None] * delta
    center, right = left + delta, left + 2.0 * delta
    mel = _mel(bin_width * np.arange(n_fft_bins, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    fb = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([fb, np.zeros((num_bins, 1))], 1).astype(np.float32)


def hanning_window(n: int) -> np.ndarray:
    """torch.hann_window(n, periodic=False)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / (n - 1))).astype(np.float32)


class _Tables:
    """Window + filterbank for one (sample_rate, num_mel_bins) pair, uploaded once per device."""

    def __init__(self):
        self._cache = {}

    def get(self, device, sample_rate: int, num_mel_bins: int):
        key = (str(device), sample_rate, num_mel_bins)
        t = self._cache.get(key)
        if t is None:
            win = int(sample_rate * 0.025)
            shift = int(sample_rate * 0.010)
            nfft = 1 << (win - 1).bit_length()                    # round_to_power_of_two
            t = (torch.from_numpy(hanning_window(win)).to(device),
                 torch.from_numpy(kaldi_mel_banks(num_mel_bins, nfft, float(sample_rate))).to(device), shift, nfft)
            self._cache[key] = t
        return t


_TABLES = _Tables()


def waveform_to_melspectrogram(waveform: Union[np.ndarray, torch.Tensor], num_mel_bins: int = 128, target_length: int = 204,
                               sample_rate: int = 16000, clip_duration: float = 2.0, mean: float = -4.268, std: float = 9.138,
                               device=None) -> torch.Tensor:
    """(channels, samples) waveform -> (1, num_mel_bins, target_length) f32 on `device` (default: cuda)."""
    if isinstance(waveform, np.ndarray):
        waveform = torch.from_numpy(waveform)
    if waveform.dim() != 2:
        raise ValueError(f"waveform must be (channels, samples), got {tuple(waveform.shape)}")
    n = waveform.shape[1]
    n_target = int(clip_duration * sample_rate)
    start = (n - n_target) // 2 if n > n_target else 0
    if device is None:
        device = waveform.device if waveform.is_cuda else torch.device("cuda")
    clip = waveform[:1, start:start + n_target].to(device=device, dtype=torch.float32).contiguous()   # kaldi.fbank: channel 0
    window, mel_fb, shift, nfft = _TABLES.get(device, sample_rate, num_mel_bins)
    return ops.kaldi_fbank(clip, window, mel_fb, shift=shift, nfft=nfft, t_out=target_length, mean=mean, std=std)


class AudioMelspectrogramExtractor:
    """Same constructor / call contract as the reference class (avgen/data/utils.py:58-110): list of (c, n) waveforms
    -> (b, 1, num_mel_bins, target_length) features."""

    def __init__(self, num_mel_bins=128, target_length=204, sample_rate=16000, clip_duration=2, mean=-4.268, std=9.138):
        self.num_mel_bins = num_mel_bins
        self.target_length = target_length
        self.sample_rate = sample_rate
        self.clip_duration = clip_duration
        self.mean = mean
        self.std = std

    @property
    def max_length_s(self) -> int:
        return self.clip_duration

    @property
    def sampling_rate(self) -> int:
        return self.sample_rate

    def __call__(self, waveforms: Union[np.ndarray, torch.Tensor, List[np.ndarray], List[torch.Tensor]], device=None) -> torch.Tensor:
        if isinstance(waveforms, (np.ndarray, torch.Tensor)) and waveforms.ndim == 2:
            waveforms = [waveforms]
        feats = [waveform_to_melspectrogram(w, self.num_mel_bins, self.target_length, self.sample_rate, self.clip_duration,
                                            self.mean, self.std, device=device) for w in waveforms]
        return torch.stack(feats, 0)      # (b, 1, n_mel, t)
