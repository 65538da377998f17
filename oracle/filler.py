"""Closed-form deterministic parameter filler shared by oracle/gen_golden.py (which fills the
REFERENCE model) and the tests / bench (which fill the oracle's and the product's state_dict), so no
weight file ever needs to be committed: a parameter's values depend only on its name and shape.

The reference zero-initialises every temporal path (conv_temp: utils.py:31-32; attn_temp.to_out:
ff_spatio_audio_temp_transformer_3d.py:267); the filler randomises them like everything else, so the
golden vectors exercise those paths.
"""
from __future__ import annotations

import math
import zlib

import torch


def fill_tensor(name: str, shape, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    if len(shape) >= 2:                       # linear / conv weight: keep activations O(1)
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
    elif name.endswith("weight"):             # norm scale
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    else:                                     # bias / norm shift
        t = 0.05 * torch.randn(shape, generator=g)
    return t.to(dtype)


def heavy_tail_scale(name: str, n_out: int) -> torch.Tensor:
    """per-output-channel gains (deterministic in the parameter name): most channels 10^u with u uniform in [-1, 0.3], and
    one channel in 16 an outlier with a gain of 10..100 — the outlier-channel regime of trained diffusion UNets, which
    closed-form N(0, 1/fan_in) weights never visit.  Gains span three decades across the channels of a layer and
    are normalised to unit RMS per matrix (un-normalised gains of 1e-2 .. 1e2 on every layer make the reference's own fp32
    forward overflow: 1.5e25 inside the tiny network)."""
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ 0x5A5A5A5A) & 0x7FFFFFFF)
    u = 1.3 * torch.rand(n_out, generator=g) - 1.0
    out = torch.rand(n_out, generator=g) < 1.0 / 16
    u = torch.where(out, 1.0 + torch.rand(n_out, generator=g), u)
    gain = 10.0 ** u
    return gain / gain.pow(2).mean().sqrt()          # unit RMS: the spread stays, the residual stream does not explode


def fill_state_dict(shapes: dict, dtype=torch.float32) -> dict:
    """shapes: name -> shape (e.g. tests/golden/unet_sd15_state_dict_shapes.json)."""
    return {k: fill_tensor(k, v, dtype) for k, v in shapes.items()}


def fill_module_(module: torch.nn.Module, prefix: str = "", round_bf16: bool = False, heavy_tail: bool = False) -> None:
    """round_bf16: matrices (dim >= 2) are rounded to bf16-representable values, so a bf16-weight implementation holds
    exactly the parameters the fp32 reference ran with (block-level goldens).  heavy_tail: every matrix gets per-output-channel
    gains spanning 1e-2 .. 1e2 (heavy_tail_scale)."""
    with torch.no_grad():
        for k, p in module.state_dict().items():
            t = fill_tensor(prefix + k, p.shape, p.dtype)
            if heavy_tail and t.dim() >= 2:
                t = t * heavy_tail_scale(prefix + k, t.shape[0]).reshape(-1, *([1] * (t.dim() - 1))).to(t.dtype)
            if round_bf16 and t.dim() >= 2:
                t = t.to(torch.bfloat16).to(p.dtype)
            p.copy_(t)


def seeded_randn(seed: int, *shape) -> torch.Tensor:
    return torch.randn(*shape, generator=torch.Generator(device="cpu").manual_seed(seed))


def seeded_randn_bf16(seed: int, *shape) -> torch.Tensor:
    """bf16-representable f32 values: both sides of a parity test read exactly the same numbers."""
    return seeded_randn(seed, *shape).to(torch.bfloat16).float()
