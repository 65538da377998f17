"""Multi-GPU path on CPU (-m "not gpu"): world_size-2 `gloo` processes exercise exactly what bench.py / the dataset driver
do with RCCL on a node — clip sharding, ONE broadcast of the packed weight blob into a layout-only (meta) replica, per-rank
independent work, ONE all-gather of metrics.  No collective sits inside the denoising loop."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import asva_amd.engine as e
    import asva_amd.unet as u
    from asva_amd import dist as adist
    from asva_amd.conditioning import audio_segment_mask
    from asva_amd.engine import DenoiseEngine
    from asva_amd.schedulers import DDIMScheduler
    from asva_amd.unet import AudioUNet3DConditionModel
    from oracle.filler import fill_module_
    from tests import emu_ops

    u.ops = emu_ops
    e.ops = emu_ops
    torch.set_num_threads(2)
    adist.init_process_group("gloo")
    assert dist.get_world_size() == world
    g = load_golden("unet_tiny_e2e.pt")
    # rank 0 owns real weights; the others build the layout from meta parameters and receive the bytes
    if rank == 0:
        unet = AudioUNet3DConditionModel.from_config(g["config"]).eval()
        fill_module_(unet)
        pk = unet.pack("cpu")
    else:
        with torch.device("meta"):
            unet = AudioUNet3DConditionModel.from_config(g["config"]).eval()
        pk = unet.pack("cpu")
        assert int(pk.blob.count_nonzero()) == 0
    adist.broadcast_blob(pk.blob, src=0)
    checksum = int(pk.blob.to(torch.int64).sum())
    # 5 clips over 2 ranks: i -> i mod world
    mine = adist.shard_clips(5, rank, world)
    eng = DenoiseEngine(unet, DDIMScheduler(), audio_guidance_scale=4.0)
    finals = {}
    # the witness clip (id 0) on EVERY rank, as bench.py does: same seed-by-clip-id inputs, weights from the broadcast
    gen = torch.Generator().manual_seed(adist.clip_seed(0, base=100))
    wlat = torch.randn(1, 4, *g["sample"].shape[2:], generator=gen)
    eng.set_conditioning(g["text"][:1], g["audio"][1:2], g["audio"][:1], audio_segment_mask(wlat.shape[2]), wlat.shape[2])
    witness = adist.bit_checksum(eng.run(wlat, 1))
    for ci in mine:
        gen = torch.Generator().manual_seed(adist.clip_seed(ci, base=100))
        lat = torch.randn(1, 4, *g["sample"].shape[2:], generator=gen)
        eng.set_conditioning(g["text"][:1], g["audio"][1:2], g["audio"][:1], audio_segment_mask(lat.shape[2]), lat.shape[2])
        finals[ci] = float(eng.run(lat, 2).double().sum())
    rows = adist.gather_metrics([float(len(mine)), float(checksum % 1000003), sum(finals.values()), witness])
    adist.barrier()
    q.put((rank, mine, checksum, finals, rows))
    dist.destroy_process_group()


def test_two_rank_sharded_denoise_with_weight_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, cs0, fin0, rows0), (r1, mine1, cs1, fin1, rows1) = res
    assert mine0 == [0, 2, 4] and mine1 == [1, 3]
    assert cs0 == cs1 and cs0 != 0                       # the meta replica holds rank 0's packed bytes after ONE broadcast
    assert rows0 == rows1 and [r[0] for r in rows0] == [3.0, 2.0]     # all-gather: every rank sees every rank's metrics
    assert rows0[0][1] == rows0[1][1]
    assert rows0[0][3] == rows0[1][3] and rows0[0][3] > 0   # both ranks computed the witness clip bit for bit (bench.py's self-check)
    assert all(map(lambda v: v == v, list(fin0.values()) + list(fin1.values())))   # finite


def test_bit_checksum_sees_one_flipped_bit():
    from asva_amd.dist import bit_checksum, clip_seed

    t = torch.randn(3, 5)
    u = t.clone()
    u.view(torch.int32)[1, 2] ^= 1
    assert bit_checksum(t) == bit_checksum(t.clone()) and bit_checksum(t) != bit_checksum(u)
    assert clip_seed(3) == 1003 and clip_seed(3, base=7) == 10


def test_shard_clips_partition():
    from asva_amd.dist import shard_clips

    for n in (0, 1, 7, 32):
        for world in (1, 2, 8):
            parts = [shard_clips(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_device_index_and_backend_switches(monkeypatch):
    """AVSD_DIST_SAME_DEVICE / AVSD_DIST_BACKEND (tests/test_multirank_gpu.py runs two ranks on the one leased GPU with them)"""
    from asva_amd import dist as adist

    monkeypatch.delenv("AVSD_DIST_SAME_DEVICE", raising=False)
    monkeypatch.delenv("AVSD_DIST_BACKEND", raising=False)
    assert adist.same_device() is False and adist.device_index(3) == 3          # no GPU visible here: nothing to check against
    assert adist.backend_name("nccl") == "nccl" and adist.backend_name() == "gloo"
    monkeypatch.setenv("AVSD_DIST_SAME_DEVICE", "1")
    assert adist.same_device() is True and adist.device_index(0) == 0 and adist.device_index(5) == 0
    monkeypatch.setenv("AVSD_DIST_BACKEND", "gloo")
    assert adist.backend_name("nccl") == "gloo"
    monkeypatch.setenv("AVSD_DIST_BACKEND", "mpi")
    with pytest.raises(ValueError):
        adist.backend_name()
    # more ranks than GPUs without the switch is refused
    monkeypatch.delenv("AVSD_DIST_SAME_DEVICE")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert adist.device_index(0) == 0
    with pytest.raises(RuntimeError):
        adist.device_index(1)
