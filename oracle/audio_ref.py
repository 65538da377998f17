"""CPU restatement of the audio conditioning front-end — TEST INFRASTRUCTURE, not product code.

Follows the reference call sites avgen/data/utils.py:26-55 (`waveform_to_melspectrogram`) and
avgen/models/audio_encoders/segmask_imagebind.py:80-123 (`ImageBindSegmaskAudioEncoder.forward`).  Everything those
sites delegate to is third-party and absent from /root/reference and from this image:
  * facebookresearch/ImageBind (git submodule `submodules/ImageBind`, README.md:61; no pinned commit in the tree):
    `imagebind.data.waveform2melspec`, `imagebind.models.imagebind_model.imagebind_huge` audio preprocessor / trunk /
    head;
  * torchaudio `compliance.kaldi.fbank` (requirements.txt: torchaudio==2.0.x).
PARITY UNPINNED: no reference-side vector exists for this row.  The restatement follows the published algorithms
(Kaldi `compute-fbank-feats` as implemented by torchaudio; ImageBind's SimpleTransformer) and is pinned where a torch
built-in implements the same arithmetic: torch.fft.rfft for the DFT, torch.nn.MultiheadAttention(add_bias_kv=True)
for the attention with the appended key/value pair, F.layer_norm, F.gelu, F.conv2d (tests/test_oracle.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

EPS32 = float(np.finfo(np.float32).eps)


# ---- Kaldi filterbank (float64 throughout; torchaudio computes the same in float32) -------------------------------
def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_banks(num_bins=128, padded=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    """torchaudio `get_mel_banks` (vtln off): triangles spaced uniformly in mel between low_freq and Nyquist, evaluated at the
    centre frequency of FFT bins 0 .. padded/2 - 1."""
    nyq = 0.5 * sample_freq
    high = high_freq + nyq if high_freq <= 0 else high_freq
    lo, hi = mel_scale(low_freq), mel_scale(high)
    delta = (hi - lo) / (num_bins + 1)
    fb = np.zeros((num_bins, padded // 2), dtype=np.float64)
    for b in range(num_bins):
        left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
        for i in range(padded // 2):
            m = mel_scale(sample_freq / padded * i)
            if left < m < right:
                fb[b, i] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
    return fb


def kaldi_fbank(wave: np.ndarray, sample_freq=16000.0, num_mel_bins=128, frame_length_ms=25.0, frame_shift_ms=10.0,
                preemph=0.97) -> np.ndarray:
    """wave (n,) -> (frames, num_mel_bins) log-mel energies: snip_edges framing, per-frame DC removal, pre-emphasis,
    hanning window, zero-pad to a power of two, power spectrum, mel filters, log with a float32-eps floor."""
    wave = np.asarray(wave, dtype=np.float64)
    win = int(sample_freq * frame_length_ms * 0.001)
    shift = int(sample_freq * frame_shift_ms * 0.001)
    padded = 1 << (win - 1).bit_length()
    if wave.shape[0] < win:
        return np.zeros((0, num_mel_bins))
    n_frames = 1 + (wave.shape[0] - win) // shift
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / (win - 1))
    fb = mel_banks(num_mel_bins, padded, sample_freq)
    out = np.zeros((n_frames, num_mel_bins))
    for t in range(n_frames):
        fr = wave[t * shift: t * shift + win].copy()
        fr -= fr.mean()
        fr = fr - preemph * np.concatenate([fr[:1], fr[:-1]])
        fr *= window
        spec = np.abs(np.fft.rfft(fr, n=padded)) ** 2                     # padded/2 + 1 bins; the Nyquist bin gets weight 0
        out[t] = np.log(np.maximum(fb @ spec[: padded // 2], EPS32))
    return out


def waveform_to_melspectrogram_ref(waveform: np.ndarray, num_mel_bins=128, target_length=204, sample_rate=16000,
                                   clip_duration=2.0, mean=-4.268, std=9.138) -> np.ndarray:
    """avgen/data/utils.py:26-55 + ImageBind waveform2melspec: (c, n) -> (1, num_mel_bins, target_length) float32."""
    waveform = np.asarray(waveform, dtype=np.float64)
    n, n_target = waveform.shape[1], int(clip_duration * sample_rate)
    start = (n - n_target) // 2 if n > n_target else 0
    clip = waveform[:, start:start + n_target]
    clip = clip - clip.mean()                                             # waveform2melspec; cancelled by the DC removal
    fb = kaldi_fbank(clip[0], float(sample_rate), num_mel_bins).T         # (mel, frames)
    p = target_length - fb.shape[1]
    fb = np.pad(fb, ((0, 0), (0, p))) if p > 0 else fb[:, :target_length]
    return ((fb - mean) / std)[None].astype(np.float32)


# ---- ImageBind-Huge audio branch + final_layer_norm --------------------------------------------------------------
def mha_bias_kv(x, in_w, in_b, bias_k, bias_v, out_w, out_b, heads):
    """torch.nn.MultiheadAttention(batch_first semantics, add_bias_kv=True, no masks): the learned pair is appended
    after the projected keys / values."""
    b, L, C = x.shape
    q, k, v = F.linear(x, in_w, in_b).chunk(3, -1)
    k = torch.cat([k, bias_k.reshape(1, 1, C).expand(b, 1, C)], 1)
    v = torch.cat([v, bias_v.reshape(1, 1, C).expand(b, 1, C)], 1)
    d = C // heads
    sp = lambda t: t.reshape(b, -1, heads, d).transpose(1, 2)             # noqa: E731
    att = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(d), -1) @ sp(v)
    return F.linear(att.transpose(1, 2).reshape(b, L, C), out_w, out_b)


def audio_encoder_ref(sd, mel: torch.Tensor, heads=12, depth=12):
    """segmask_imagebind.py:80-101 on a state dict with ImageBind's names: -> (cls_embeds (b, 1024), encodings (b, 229, 768))."""
    mel = mel.float()
    x = F.conv2d(mel, sd["preprocessor.rgbt_stem.proj.weight"].float(), None, stride=10)                # (b, 768, 12, 19)
    x = x.flatten(2).transpose(1, 2)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd["preprocessor.rgbt_stem.norm_layer.weight"].float(), sd["preprocessor.rgbt_stem.norm_layer.bias"].float(), 1e-5)
    x = torch.cat([sd["preprocessor.cls_token"].float().expand(x.shape[0], -1, -1), x], 1)
    x = x + sd["preprocessor.pos_embedding_helper.pos_embed"].float()
    for i in range(depth):
        p = f"trunk.blocks.{i}."
        g = lambda k: sd[p + k].float()                                    # noqa: E731
        h = F.layer_norm(x, (C,), g("norm_1.weight"), g("norm_1.bias"), 1e-6)
        x = x + mha_bias_kv(h, g("attn.in_proj_weight"), g("attn.in_proj_bias"), g("attn.bias_k"), g("attn.bias_v"),
                            g("attn.out_proj.weight"), g("attn.out_proj.bias"), heads)
        h = F.layer_norm(x, (C,), g("norm_2.weight"), g("norm_2.bias"), 1e-6)
        x = x + F.linear(F.gelu(F.linear(h, g("mlp.fc1.weight"), g("mlp.fc1.bias"))), g("mlp.fc2.weight"), g("mlp.fc2.bias"))
    cls = F.layer_norm(x, (C,), sd["head.0.weight"].float(), sd["head.0.bias"].float(), 1e-6)[:, 0]
    cls = F.linear(cls, sd["head.2.weight"].float())
    enc = F.layer_norm(x, (C,), sd["final_layer_norm.weight"].float(), sd["final_layer_norm.bias"].float(), 1e-6)
    return cls, enc
