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


def same_device() -> bool:
    """AVSD_DIST_SAME_DEVICE=1: every rank of the job uses cuda:0.  A test switch for boxes with fewer GPUs than ranks (the
    round-end GPU lease is ONE MI355X): the whole N > 1 code path — launcher, rendezvous, clip sharding, weight broadcast into
    a layout-only replica, per-rank engines, metric all-gather, witness check — then runs for real, two processes sharing one
    GPU.  RCCL refuses two ranks on one device, so the switch goes with AVSD_DIST_BACKEND=gloo."""
    return os.environ.get("AVSD_DIST_SAME_DEVICE") == "1"


def device_index(local_rank: int) -> int:
    """the GPU of a rank: cuda:LOCAL_RANK (one process per GPU), or cuda:0 for all under the same-device switch.  Anything
    else — more ranks than visible GPUs without the switch — is refused instead of silently stacking ranks on a device."""
    if same_device():
        return 0
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n and local_rank >= n:
        raise RuntimeError(f"LOCAL_RANK {local_rank} but only {n} GPU(s) are visible (AVSD_DIST_SAME_DEVICE=1 + "
                           "AVSD_DIST_BACKEND=gloo share one GPU between ranks, for tests)")
    return local_rank


def backend_name(default: str | None = None) -> str:
    """collective backend: AVSD_DIST_BACKEND overrides; "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU"""
    env = os.environ.get("AVSD_DIST_BACKEND")
    if env:
        if env not in ("nccl", "gloo"):
            raise ValueError(f"AVSD_DIST_BACKEND={env!r}: expected nccl or gloo")
        return env
    if default is not None:
        return default
    return "nccl" if torch.cuda.is_available() else "gloo"


def init_process_group(backend: str | None = None) -> None:
    rank, local_rank, world = env_rank_world()
    force = os.environ.get("AVSD_FORCE_DIST") == "1"      # lets a 1-GPU box exercise the RCCL calls (world size 1)
    if (world <= 1 and not force) or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    backend = backend_name(backend)
    if backend == "nccl":
        if same_device() and world > 1:
            raise RuntimeError("AVSD_DIST_SAME_DEVICE=1 needs AVSD_DIST_BACKEND=gloo: RCCL cannot place two ranks on one GPU")
        torch.cuda.set_device(device_index(local_rank))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)


def _host_staged(t: torch.Tensor) -> bool:
    """gloo moves host memory: device tensors take one staging copy each way (test configurations only; RCCL moves
    device memory over xGMI directly)"""
    return dist.is_initialized() and dist.get_backend() == "gloo" and t.is_cuda


def shard_clips(num_clips: int, rank: int, world: int) -> List[int]:
    """clip i -> rank i mod world."""
    return list(range(rank, num_clips, world))


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """One collective for all weights: the packed uint8 blob (UNet ~2.4 GB bf16) from `src` to every rank.
    xGMI is point-to-point, so RCCL runs this as a ring/tree at per-link rate; it happens once, outside the
    denoising loop."""
    if dist.is_initialized():
        if _host_staged(blob):
            host = blob.cpu()
            dist.broadcast(host, src=src)
            if dist.get_rank() != src:
                blob.copy_(host)
        else:
            dist.broadcast(blob, src=src)
    return blob


def gather_metrics(values: Sequence[float], device=None) -> List[List[float]]:
    """all-gather of a few floats per rank (clips, steps, seconds, ...)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return [t.tolist()]
    if _host_staged(t):
        t = t.cpu()
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
