"""RCCL on the GPU box (-m gpu).  The round-end driver leases ONE MI355X for the GPU tests, so what can be proven on real
RCCL there is the single-rank form of the two collectives of the path (one broadcast of the packed weight blob, one
all-gather of metrics) — AVSD_FORCE_DIST=1 makes a world of one initialise the `nccl` (= RCCL) communicator — plus the fill of a
layout-only (meta) replica from the broadcast buffer, which is what ranks 1.. do.  The world-size-2 logic runs on gloo in
tests/test_dist_cpu.py; the 1 -> 8 curve is the driver's to measure."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_its_collectives_on_rccl_with_one_rank():
    env = dict(os.environ, AVSD_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-vae", "--also-clips", "0", "--no-roofline", "--no-precise"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["world_size"] == 1 and out["n_gpus"] == 1 and out["all_finite"] and out["value"] > 0


def test_broadcast_fills_a_meta_replica_and_all_gather_returns_the_row():
    code = r"""
import os, torch, torch.distributed as dist
from asva_amd import dist as adist
from asva_amd.unet import AudioUNet3DConditionModel
from tests.helpers import load_golden, filled_unet
adist.init_process_group("nccl")
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
g = load_golden("unet_tiny_e2e.pt")
src = filled_unet(g["config"]).to("cuda:0")
blob0 = src.pack().blob
with torch.device("meta"):
    rep = AudioUNet3DConditionModel.from_config(g["config"]).eval()
pk = rep.pack(torch.device("cuda", 0))                       # layout only: zeros
assert int(pk.blob.count_nonzero()) == 0 and pk.blob.numel() == blob0.numel()
pk.blob.copy_(blob0)                                          # what the broadcast delivers on ranks 1..
adist.broadcast_blob(pk.blob, src=0)                          # ... and the collective itself, on RCCL
assert torch.equal(pk.blob, blob0)
B, Fr = g["sample"].shape[0], g["sample"].shape[2]
text = g["text"][:, None].expand(B, Fr, *g["text"].shape[1:]).cuda()
audio = g["audio"][:, None].expand(B, Fr, *g["audio"].shape[1:]).cuda()
mask = g["mask"][None].expand(B, -1, -1)
a = src(g["sample"].cuda(), 981, text, audio, audio_attention_mask=mask).sample
b = rep(g["sample"].cuda(), 981, text, audio, audio_attention_mask=mask).sample
assert torch.equal(a, b)                                      # the replica computes exactly what rank 0 does
rows = adist.gather_metrics([3.0, 1.5, 42.0], device=torch.device("cuda", 0))
assert rows == [[3.0, 1.5, 42.0]]
adist.barrier()
dist.destroy_process_group()
print("RCCL-OK")
"""
    env = dict(os.environ, AVSD_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
