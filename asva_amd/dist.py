"""Data-parallel sharding of independent clips over the GPUs of one node (one process per GPU).

Each (image, audio) clip is an independent denoising trajectory (the reference runs them one by one:
pipeline_audio_cond_animation.py:432-447, 532-551), so the path shards with NO collective inside the loop.
RCCL (torch.distributed backend "nccl" on ROCm) over xGMI is used exactly twice: one broadcast of the
packed weight blob at start-up and one all-gather of per-rank metrics at the end.  The same code runs on
`gloo` for the world_size-2 CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: str | None = None) -> None:
    rank, local_rank, world = env_rank_world()
    force = os.environ.get("AVSD_FORCE_DIST") == "1"      # lets a 1-GPU box exercise the RCCL calls (world size 1)
    if (world <= 1 and not force) or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)


def shard_clips(num_clips: int, rank: int, world: int) -> List[int]:
    """clip i -> rank i mod world."""
    return list(range(rank, num_clips, world))


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """One collective for all weights: the packed uint8 blob (UNet ~2.4 GB bf16) from `src` to every rank.
    xGMI is point-to-point, so RCCL runs this as a ring/tree at per-link rate; it happens once, outside the
    denoising loop."""
    if dist.is_initialized():
        dist.broadcast(blob, src=src)
    return blob


def gather_metrics(values: Sequence[float], device=None) -> List[List[float]]:
    """all-gather of a few floats per rank (clips, steps, seconds, ...)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def bit_checksum(t: torch.Tensor) -> float:
    """order-independent checksum of the BITS of a tensor (sum of its 32-bit words mod 2^52, exact in the float64 row of
    gather_metrics): two ranks that computed the same clip bit for bit report the same number."""
    w = t.detach().contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return float(int(w.sum().item()) & ((1 << 52) - 1))


def clip_seed(clip_id: int, base: int = 1000) -> int:
    """inputs are seeded by CLIP id, never by rank (the reference seeds per clip: pipeline_audio_cond_animation.py:431-447), so a
    clip gives the same result whichever rank it lands on"""
    return base + int(clip_id)


def barrier():
    if dist.is_initialized():
        dist.barrier()
