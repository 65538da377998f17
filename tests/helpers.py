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
    fill_module_(m, heavy_tail=heavy_tail)
    return m.to(dtype)


def bf16_round_state_dict(sd):
    return {k: v.to(torch.bfloat16).float() if v.dim() >= 2 else v.float() for k, v in sd.items()}
