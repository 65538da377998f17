"""Shared test helpers: golden loading, filler-weight models, error metrics."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)


def load_shapes(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def filled_unet(cfg, dtype=torch.float32, heavy_tail=False):
    """Product model with the closed-form filler weights (oracle/filler.py) — the same weights the
    reference model had when the golden vectors were generated."""
    from asva_amd.unet import AudioUNet3DConditionModel
    from oracle.filler import fill_module_

    m = AudioUNet3DConditionModel.from_config(cfg).eval()
    # the SD1.5-shaped model is filled a dozen times per session (1.17 B values from a single-threaded generator, ~5 s each):
    # keep the last filled state_dict of a configuration and copy it
    key = (json.dumps(dict(cfg), sort_keys=True, default=str), bool(heavy_tail))
    if _FILLED.get("key") == key:
        m.load_state_dict(_FILLED["sd"])
    else:
        fill_module_(m, heavy_tail=heavy_tail)
        if sum(p.numel() for p in m.parameters()) > 100_000_000:
            _FILLED.clear()
            _FILLED.update(key=key, sd={k: v.detach().clone() for k, v in m.state_dict().items()})
    return m.to(dtype)


_FILLED: dict = {}


def bf16_round_state_dict(sd):
    return {k: v.to(torch.bfloat16).float() if v.dim() >= 2 else v.float() for k, v in sd.items()}
