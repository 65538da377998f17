"""Closed-form deterministic parameter filler: a parameter's values depend only on its name and shape, so a model with
"filler weights" can be rebuilt anywhere without a weight file.  The golden vectors under tests/golden/ come from the
REFERENCE model filled by the same rule (oracle/filler.py is the reference-side statement of it; tests/test_host_cpu.py
asserts the two produce identical bits).  bench.py uses this side to measure the split-precision mode's rel-L2 against
tests/golden/unet_sd15_forward.pt inside the bench run."""
from __future__ import annotations

import math
import zlib

import torch


def fill_tensor(name: str, shape, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    if len(shape) >= 2:                       # matrix: N(0, 1 / fan_in)
        fan_in = math.prod(shape[1:])
        t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
    elif name.endswith("weight"):             # norm gain around 1
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    else:                                     # bias / norm shift
        t = 0.05 * torch.randn(shape, generator=g)
    return t.to(dtype)


def fill_module_(module: torch.nn.Module, prefix: str = "") -> None:
    with torch.no_grad():
        for k, p in module.state_dict().items():
            p.copy_(fill_tensor(prefix + k, p.shape, p.dtype))


def seeded_randn(seed: int, *shape) -> torch.Tensor:
    return torch.randn(*shape, generator=torch.Generator(device="cpu").manual_seed(seed))
