"""Storage precision of activations and matrix weights (accumulation is always f32).

  "bf16" (default)  bfloat16 storage, v_mfma_f32_32x32x16_bf16 — what BASELINE.json's metric is quoted in.
  "fp16"            IEEE half storage, v_mfma_f32_32x32x16_f16 at the same MFMA rate: 3 more mantissa bits (rounding
                    2^-12 instead of 2^-9 per element), range 6e-5 .. 65504.  The reference's own dataset driver defaults to
                    torch.float16 (pipeline_audio_cond_animation.py:485).  Served by libavsd_hip_f16.so = the same kernel
                    sources compiled with -DAVSD_F16=1 (asva_amd/build.py).

`set_precision` switches the whole process: models must be (re)packed after a switch (pack() keys its cache on it).
"""
from __future__ import annotations

import os

import torch

_NAMES = {"bf16": torch.bfloat16, "fp16": torch.float16}
NAME = "bf16"
ACT = torch.bfloat16


def set_precision(name: str) -> None:
    global NAME, ACT
    if name not in _NAMES:
        raise ValueError(f"precision must be one of {sorted(_NAMES)}, got {name!r}")
    NAME, ACT = name, _NAMES[name]


if os.environ.get("AVSD_PRECISION"):
    set_precision(os.environ["AVSD_PRECISION"])
