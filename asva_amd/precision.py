"""Storage precision of activations and matrix weights (accumulation is always f32).

  "bf16" (default)  bfloat16 storage, v_mfma_f32_32x32x16_bf16 — what BASELINE.json's metric is quoted in.
  "fp16"            IEEE half storage, v_mfma_f32_32x32x16_f16 at the same MFMA rate: 3 more mantissa bits (rounding
                    2^-12 instead of 2^-9 per element), range 6e-5 .. 65504.  The reference's own dataset driver defaults to
                    torch.float16 (pipeline_audio_cond_animation.py:485).  Served by libavsd_hip_f16.so = the same kernel
                    sources compiled with -DAVSD_F16=1 (asva_amd/build.py).

`set_precision` switches the whole process: models must be (re)packed after a switch (pack() keys its cache on it).

Split precision ("x2", `set_split(True)`; the mode that meets BASELINE.json's 1e-3 against the reference's fp32 pipeline,
scripts/animation_gen.py:43-44): every 16-bit tensor — activations and matrix weights alike — becomes a PAIR of planes,
main = round16(v) and rest = round16(v - main), i.e. 16 significant bits in bf16 with f32's range.  Matrix products run as
three MFMA passes (main.main + rest.main + main.rest) into the same f32 accumulator, everything else reconstructs main + rest
in f32.  Storage convention on the host: a split tensor is a view into the FIRST half of its storage and its rest plane sits
at the same offset in the second half (ops.alloc16 / weights.to_act / the packed blob are laid out that way), so views and
slices carry their rest plane along and the kernels get it as one element offset.
"""
from __future__ import annotations

import os

import torch

_NAMES = {"bf16": torch.bfloat16, "fp16": torch.float16}
NAME = "bf16"
ACT = torch.bfloat16


SPLIT = False


def set_split(on: bool) -> None:
    """switches split-precision storage on / off for the whole process (models are repacked on next use)"""
    global SPLIT
    SPLIT = bool(on)


def set_precision(name: str) -> None:
    global NAME, ACT
    if name not in _NAMES:
        raise ValueError(f"precision must be one of {sorted(_NAMES)}, got {name!r}")
    NAME, ACT = name, _NAMES[name]


if os.environ.get("AVSD_PRECISION"):
    set_precision(os.environ["AVSD_PRECISION"])
if os.environ.get("AVSD_SPLIT", "0") != "0":
    set_split(True)
