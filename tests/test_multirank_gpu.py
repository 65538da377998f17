"""The N > 1 code path executed for real on the one-GPU lease (-m gpu): two processes started by torch.distributed.run, both
on cuda:0 (AVSD_DIST_SAME_DEVICE=1), the two collectives of the path (one weight broadcast, one metric all-gather)
on gloo (AVSD_DIST_BACKEND=gloo — RCCL refuses two ranks on one device).  Everything else is what an 8-GPU node runs:
bench.py's self-launch, rendezvous on 127.0.0.1, rank 1 building a layout-only replica and receiving the 2.4-GB blob, one
clip per rank, the witness-clip bit checksum across ranks, per-rank rates; and the reference's dataset driver sharding its
video list by rank (pipeline_audio_cond_animation.py:532-551 loops sequentially)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AVSD_FORCE_DIST"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_bench_two_ranks_on_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                        "--no-roofline", "--no-precise", "--also-clips", "0"],
                       env=_env(AVSD_DIST_SAME_DEVICE="1", AVSD_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                              # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["dist_backend"] == "gloo" and out["ranks_share_one_gpu"] is True
    assert out["ranks_agree_on_witness_clip"] is True and out["all_finite"] is True
    assert len(out["per_rank_steps_per_s"]) == 2 and all(v > 0 for v in out["per_rank_steps_per_s"])
    assert out["scaling"] == "weak" and out["config"]["parallelism"].startswith("dp2")
    # whole-job value = all ranks' steps / the slowest rank's wall time
    assert out["value"] == pytest.approx(2 * 6 / (out["ms_per_step"] * 6e-3), rel=1e-3)
    assert out["vae_decode"]["clips_per_s"] > 0


def test_bench_refuses_more_ranks_than_gpus_without_the_switch():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with one GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "only 1 GPU" in (r.stderr + r.stdout)
    # ... and the switch without gloo (two RCCL ranks on one device) is refused by the rank processes, not attempted
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(AVSD_DIST_SAME_DEVICE="1"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "AVSD_DIST_BACKEND=gloo" in (r.stderr + r.stdout)


def test_dataset_driver_two_ranks_disjoint_clip_sets(tmp_path):
    from tests.test_dataset_driver_gpu import FPS, FRAMES, NCLIPS, SIZE, STEPS, _write_tree

    names = _write_tree(tmp_path)
    # a third video so that the shards are uneven: rank 0 -> videos 0, 2; rank 1 -> video 1
    ds = tmp_path / "datasets" / "AVSync15"
    import shutil

    shutil.copy(ds / "videos" / names[0], ds / "videos" / "dog" / "clip_c.npz")
    names = names + ["dog/clip_c.npz"]
    (ds / "test.txt").write_text("\n".join(names) + "\n")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "tests", "multirank_worker.py"), str(STEPS), str(FRAMES), str(FPS), str(SIZE[0]), str(NCLIPS)]
    r = subprocess.run(cmd, env=_env(AVSD_DIST_SAME_DEVICE="1", AVSD_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rows = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [x["world"] for x in rows] == [2, 2] and [x["device"] for x in rows] == [0, 0]
    sets = [{w["file"] for w in x["written"]} for x in rows]
    out_root = os.path.join("exp", "evaluations", "checkpoint-7", "AG-4.0_TG-1.0", "seed-0", "videos")
    want = {os.path.join(out_root, n[:-4] + f"_clip-{k:02d}.mp4") for n in names for k in range(NCLIPS)}
    assert sets[0].isdisjoint(sets[1]) and (sets[0] | sets[1]) == want            # disjoint shards, union = every clip
    assert len(sets[0]) == 2 * NCLIPS and len(sets[1]) == NCLIPS                   # clip i -> rank i mod 2
    for f in want:
        stem = tmp_path / f[:-4]
        assert any(stem.with_suffix(s).is_file() for s in (".mp4", ".avi")), f
    assert all(w["shape"] == [FRAMES, SIZE[0], SIZE[1], 3] for x in rows for w in x["written"])
