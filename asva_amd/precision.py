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


# ---- per-layer precision plan (round 5) --------------------------------------------------------------------------------------
# Split precision pays three MFMA passes and two planes for EVERY tensor; its 2.6e-5 leaves 40x of BASELINE.json's 1e-3 unspent.
# oracle/precision_sensitivity.py measured where the fp16 modes lose their accuracy (profiles/r5_precision_sensitivity.json): of the
# 1.46e-3 of "fp16 + f32 residual stream", 1.33e-3 comes from ~40 cheap products that sit ON the residual path — the 1x1 shortcut
# convolutions of the ResBlocks and their temporal mixes, the down / up samplers, conv_in and conv_out; everything else (the 3x3
# convolutions inside the ResBlocks, every transformer projection, the attentions) adds up to ~5e-4.  The plan
# (asva_amd/precision_plan.json) therefore runs the whole network in IEEE-half storage with the f32 residual stream — the fast
# one-pass kernels, LDS-resident convolution and fused cross-attention block included — and only the listed product kinds as
# three-pass split products: their A operand is split into (main, rest) planes from the f32 master of the stream, their weights are
# packed as two planes, their result is written in f32.
PLAN = None
_BEFORE_PLAN = None       # (NAME, SPLIT) at the time the plan was switched on: what set_plan(False) returns to
PLAN_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "precision_plan.json")


def set_plan(on=True, path=None) -> None:
    """switches the per-layer precision plan on (fp16 storage + f32 residual stream + the plan's three-pass products) or off
    (back to the storage mode that was active before it); models are repacked on next use"""
    global PLAN, _BEFORE_PLAN
    if not on:
        if PLAN is not None and _BEFORE_PLAN is not None:      # back to the storage mode that was active when the plan was switched on
            set_precision(_BEFORE_PLAN[0])
            set_split(_BEFORE_PLAN[1])
        PLAN = None
        _BEFORE_PLAN = None
        return
    import json

    with open(path or PLAN_PATH) as f:
        plan = json.load(f)
    if plan.get("storage") != "fp16" or plan.get("residual_stream") != "f32":
        raise ValueError("precision plan: storage must be fp16 with an f32 residual stream")
    # entries: a kind (every product of that kind) or "kind@block" for one block's ("sampler@up_blocks.2", "shortcut_temp@down_blocks.1")
    unknown = {e.split("@")[0] for e in plan["three_pass"]} - {"conv_in", "conv_out", "shortcut", "shortcut_temp", "sampler", "sampler_temp"}
    if unknown:
        raise ValueError(f"precision plan: three-pass kinds {sorted(unknown)} are not built (asva_amd/unet.py _ffconv)")
    if PLAN is None:
        _BEFORE_PLAN = (NAME, SPLIT)
    set_split(False)
    set_precision("fp16")
    PLAN = {"three_pass": frozenset(plan["three_pass"]), "name": plan.get("name", "plan")}


def three_pass(kind, where=None) -> bool:
    """is this product kind (optionally: of block `where`, e.g. "up_blocks.2") a three-pass split product under the active plan?"""
    return PLAN is not None and (kind in PLAN["three_pass"] or (where is not None and f"{kind}@{where}" in PLAN["three_pass"]))


def plan_key():
    return None if PLAN is None else tuple(sorted(PLAN["three_pass"]))


if os.environ.get("AVSD_PRECISION"):
    set_precision(os.environ["AVSD_PRECISION"])
if os.environ.get("AVSD_SPLIT", "0") != "0":
    set_split(True)
if os.environ.get("AVSD_PRECISION_PLAN", "0") != "0":
    set_plan(True)
