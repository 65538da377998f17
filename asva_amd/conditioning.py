"""Conditioning contract of the denoising path (integer/shape logic only).

The frozen ImageBind audio encoder itself is out of scope; what the UNet
consumes from it is (a) 229 tokens of width 768 per clip and (b) a per-frame boolean key mask.
This module restates (b): avgen/models/audio_encoders/segmask_imagebind.py:62-78 (`_auto_split`)
and :104-114 (mask assembly): key 0 (CLS) is always visible; the 12 x 19 (frequency x time) patch
grid is visible to frame s only inside its time chunk.
"""
from __future__ import annotations

import math

import numpy as np
import torch

AUDIO_FREQ_PATCHES = 12   # "n, t = 12, 19 # hard code" (segmask_imagebind.py:104)
AUDIO_TIME_PATCHES = 19
AUDIO_TOKENS = 1 + AUDIO_FREQ_PATCHES * AUDIO_TIME_PATCHES  # 229


def auto_split(n: int, n_chunk: int) -> torch.Tensor:
    """[n_chunk, n] bool: chunk c covers `ceil(n / n_chunk)` consecutive positions starting at
    round(linspace(0, n - size, n_chunk))[c]  (numpy rounding, half to even)."""
    size = int(math.ceil(n / n_chunk))
    assert size >= 1
    starts = np.round(np.linspace(0, n - size, n_chunk, endpoint=True)).astype(np.int32)
    mask = torch.zeros(n_chunk, n, dtype=torch.bool)
    for c, s in enumerate(starts):
        mask[c, int(s): int(s) + size] = True
    return mask


def audio_segment_mask(n_segment: int) -> torch.Tensor:
    """[n_segment, 229] bool, the `audio_segment_masks` row pattern for one clip."""
    seg = auto_split(AUDIO_TIME_PATCHES, n_segment)                      # [s, t]
    seg = seg[:, None, :].expand(n_segment, AUDIO_FREQ_PATCHES, AUDIO_TIME_PATCHES).reshape(n_segment, -1)
    return torch.cat([torch.ones(n_segment, 1, dtype=torch.bool), seg], dim=1).contiguous()


def mask_to_key_index(mask: torch.Tensor) -> torch.Tensor:
    """Bool mask [frames, keys] with the same number of visible keys in every row -> int32
    [frames, visible] list of visible key indices (ascending).  Masked keys get -inf before the
    softmax in the reference, i.e. weight exactly 0, so attending to the gathered keys is identical."""
    counts = mask.sum(dim=1)
    if not bool((counts == counts[0]).all()):
        raise ValueError("mask rows expose different numbers of keys")
    idx = torch.nonzero(mask, as_tuple=False)[:, 1].reshape(mask.shape[0], int(counts[0]))
    return idx.to(torch.int32).contiguous()
