"""Launch plans (include/avsd.h "launch plans", asva_amd/plan.py) on the MI355X (-m gpu).

The Python host records `set_conditioning`, one UNet forward of the CFG batch and the VAE clip decode as plans, and
    1. replays them in-process through the C API against FRESH, zero-filled buffers (only the bundle's CONST data and the
       declared inputs are uploaded): bit-identical outputs prove the plans are complete — nothing the step needs was
       produced by a torch op the recorder cannot see;
    2. hands the bundle to tools/plan_host.cpp, a host with no Python and no torch (HIP runtime + the C ABI only), which
       runs conditioning, a whole PLMS / DDIM denoising loop and the decode: its latents and frames must equal the
       DenoiseEngine's bit for bit.
"""
import os
import subprocess

import numpy as np
import pytest
import torch

from tests.helpers import filled_unet, load_golden
from tests.test_host_cpu import TINY_VAE, _filled_vae

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def fp16_library():
    from asva_amd import precision as P

    P.set_precision("fp16")
    yield
    P.set_precision("bf16")


@pytest.fixture
def split_precision():
    from asva_amd import precision as P

    P.set_split(True)
    yield
    P.set_split(False)


def _setup(kind):
    from asva_amd import precision as P
    from asva_amd.engine import DenoiseEngine
    from asva_amd.schedulers import DDIMScheduler, PNDMScheduler

    g = load_golden("unet_tiny_e2e.pt")
    dev = torch.device("cuda", 0)
    unet, vae = filled_unet(g["config"]).to(dev), _filled_vae(TINY_VAE).to(dev)
    f, h, w = g["sample"].shape[2:]
    gen = torch.Generator().manual_seed(3)
    lat0 = torch.randn(1, 4, f, h, w, generator=gen).to(dev)
    # the CFG batch of audio-only guidance, already in the storage type: text [t, t], audio [null, a] (pipeline :150-194)
    # (split precision: the host hands f32 and the conditioning plan's first launches split it into the two planes)
    cond = torch.float32 if P.SPLIT else P.ACT
    text = torch.cat([g["text"][:1], g["text"][:1]]).to(dev, cond).contiguous()
    audio = torch.cat([g["audio"][:1], g["audio"][1:2]]).to(dev, cond).contiguous()
    eng = DenoiseEngine(unet, PNDMScheduler() if kind == "pndm" else DDIMScheduler(), audio_guidance_scale=4.0, use_graph=False)
    return g, unet, vae, eng, lat0, text, audio, f


def _record(tmp_path, kind, steps=4):
    from asva_amd import plan
    from asva_amd import unet as U

    g, unet, vae, eng, lat0, text, audio, f = _setup(kind)
    dev = lat0.device
    # everything the replaying host writes (weights, tables, inputs) exists before the first recording starts
    x, t, latents = lat0.clone(), torch.full((1,), 501.0, device=dev), lat0.clone()
    rec = plan.Recorder()
    rec.region("unet_weights", unet.pack().blob, plan.CONST)
    rec.region("vae_weights", vae.pack().blob, plan.CONST)
    rec.region("frame_index", U.frame_index(f, dev), plan.CONST)          # torch-made tables the conditioning launches read
    rec.region("audio_key_index", U.key_index_for(g["mask"].cpu().bool(), dev), plan.CONST)
    for name, tns in (("text", text), ("audio", audio), ("x", x), ("t", t), ("latents", latents)):
        rec.region(name, tns, plan.INPUT)
    with rec.record("set_conditioning"):
        unet.set_conditioning(text, audio, g["mask"], f)
    assert unet._cond.key_index is U.key_index_for(g["mask"].cpu().bool(), dev)
    with rec.record("forward"):
        noise = unet.denoise_forward(x, t, rep=2)
    rec.region("noise_pred", noise, plan.OUTPUT)
    with rec.record("decode"):
        frames = vae.decode_to_uint8_frames(latents)
    rec.region("frames", frames, plan.OUTPUT)
    path = str(tmp_path / "tiny.plan")
    bundle = rec.save(path)
    eng.prepare(lat0, steps)
    plan.export_steps(str(tmp_path / "steps.bin"), eng._ts.tolist(), eng._plans)
    return dict(g=g, unet=unet, vae=vae, eng=eng, lat0=lat0, text=text, audio=audio, noise=noise.clone(), frames=frames.clone(),
                bundle=bundle, path=path, steps=steps, t=t, shape=tuple(lat0.shape))


def _bytes(t):
    return t.detach().contiguous().reshape(-1).view(torch.uint8)


def test_plans_replay_bit_identically_from_fresh_buffers(tmp_path):
    r = _record(tmp_path, "pndm")
    b = r["bundle"]
    regs = b.regions()
    assert {"unet_weights", "vae_weights", "text", "audio", "x", "t", "noise_pred", "latents", "frames"} <= set(regs)
    print("bundle:", b.n_calls, "launches;", len(b.buffer_sizes()), f"buffers, {sum(b.buffer_sizes()) / 1e6:.1f} MB;", len(regs), "regions")
    assert b.n_calls["forward"] > 100 and b.n_calls["set_conditioning"] > 10 and b.n_calls["decode"] > 20
    b.bind_fresh(r["lat0"].device)          # new allocations, zero-filled; CONST regions loaded from tiny.plan.d/
    for name, src in (("text", r["text"]), ("audio", r["audio"]), ("x", r["lat0"]), ("t", r["t"]), ("latents", r["lat0"])):
        b.view(name).copy_(_bytes(src))
    b.run("set_conditioning")
    b.run("forward")
    b.run("decode")
    torch.cuda.synchronize()
    assert torch.equal(b.view("noise_pred"), _bytes(r["noise"])), "replayed UNet forward differs from the recording run"
    assert torch.equal(b.view("frames"), _bytes(r["frames"])), "replayed VAE decode differs from the recording run"
    # the operation-level calls (avsd_unet_set_conditioning / avsd_unet_forward / avsd_vae_decode) on a second fresh binding:
    # inputs and outputs are plain device pointers of the caller
    from asva_amd import _lib

    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    b4 = type(b)(r["path"])
    b4.bind_fresh(r["lat0"].device)
    noise_out, frames_out = torch.zeros_like(r["noise"]), torch.zeros_like(r["frames"])
    _lib.check(L.avsd_unet_set_conditioning(b4._h, r["text"].data_ptr(), r["audio"].data_ptr(), st), "avsd_unet_set_conditioning")
    _lib.check(L.avsd_unet_forward(b4._h, r["lat0"].data_ptr(), r["t"].data_ptr(), noise_out.data_ptr(), st), "avsd_unet_forward")
    _lib.check(L.avsd_vae_decode(b4._h, r["lat0"].data_ptr(), frames_out.data_ptr(), st), "avsd_vae_decode")
    torch.cuda.synchronize()
    assert torch.equal(noise_out, r["noise"]) and torch.equal(frames_out, r["frames"])
    nb = __import__("ctypes").c_int64()
    assert L.avsd_plan_region_ptr(b4._h, b"noise_pred", nb) and nb.value == r["noise"].numel() * 4
    assert not L.avsd_plan_region_ptr(b4._h, b"no_such_region", nb)
    b4.close()
    # an unbound buffer is an error, not a wild launch
    b2 = type(b)(r["path"])
    with pytest.raises(Exception, match="not bound"):
        b2.run("forward")
    b2.close()
    b.close()
    # a CONST / INPUT tensor allocated after recording started is refused: its memory may have served a temporary
    from asva_amd import plan

    rec = plan.Recorder()
    with rec.record("p"):
        r["unet"].denoise_forward(r["lat0"], r["t"], rep=2)
    rec.region("late", torch.zeros(64, device="cuda"), plan.INPUT)
    with pytest.raises(RuntimeError, match="before the first record"):
        rec.save(str(tmp_path / "late.plan"))
    # without the torch-made tables the replay must NOT match: the check above really starts from empty memory
    b3 = type(b)(r["path"])
    b3.bind_fresh(r["lat0"].device)
    b3.view("frame_index").zero_()
    for name, src in (("text", r["text"]), ("audio", r["audio"]), ("x", r["lat0"]), ("t", r["t"])):
        b3.view(name).copy_(_bytes(src))
    b3.run("set_conditioning")
    b3.run("forward")
    torch.cuda.synchronize()
    assert not torch.equal(b3.view("noise_pred"), _bytes(r["noise"]))
    b3.close()


def test_split_precision_plans_replay_bit_identically(tmp_path, split_precision):
    """The split-precision step is the same kind of launch list: twin weight blobs travel as CONST regions (both planes), the
    f32 conditioning inputs are split by recorded launches, and a replay from zero-filled buffers reproduces the recording
    run bit for bit."""
    r = _record(tmp_path, "ddim", steps=2)
    b = r["bundle"]
    b.bind_fresh(r["lat0"].device)
    for name, src in (("text", r["text"]), ("audio", r["audio"]), ("x", r["lat0"]), ("t", r["t"]), ("latents", r["lat0"])):
        b.view(name).copy_(_bytes(src))
    for p in ("set_conditioning", "forward", "decode"):
        b.run(p)
    torch.cuda.synchronize()
    assert torch.equal(b.view("noise_pred"), _bytes(r["noise"]))
    assert torch.equal(b.view("frames"), _bytes(r["frames"]))
    b.close()


@pytest.fixture
def precision_plan():
    from asva_amd import precision as P

    P.set_plan(True)
    yield
    P.set_plan(False)


def test_precision_plan_launch_plans_replay_bit_identically(tmp_path, precision_plan):
    """The per-layer precision plan (asva_amd/precision_plan.json: fp16 + f32 residual stream, three-pass split products on the residual path)
    is again only library launches — avsd_split_f32 makes the operand planes, the two-plane weights travel inside the CONST blob, the
    three-pass products are avsd_gemm_bf16 descriptors with AVSD_GEMM_X2: a bundle recorded in the mode replays from zero-filled buffers bit
    for bit."""
    r = _record(tmp_path, "ddim", steps=2)
    b = r["bundle"]
    b.bind_fresh(r["lat0"].device)
    for name, src in (("text", r["text"]), ("audio", r["audio"]), ("x", r["lat0"]), ("t", r["t"]), ("latents", r["lat0"])):
        b.view(name).copy_(_bytes(src))
    for p in ("set_conditioning", "forward", "decode"):
        b.run(p)
    torch.cuda.synchronize()
    assert torch.equal(b.view("noise_pred"), _bytes(r["noise"]))
    assert torch.equal(b.view("frames"), _bytes(r["frames"]))
    b.close()


@pytest.mark.parametrize("kind", ["pndm", "ddim"])
def test_cpp_host_runs_the_denoising_loop_without_python(tmp_path, kind):
    from asva_amd import _lib, build

    host = build.PLAN_HOST
    assert os.path.exists(host), "asva_amd/plan_host not built (python -m asva_amd.build)"
    r = _record(tmp_path, kind, steps=5)
    eng, lat0 = r["eng"], r["lat0"]
    # what the Python host computes: engine loop (eager; the graph replays the same launches) + decode
    eng.set_conditioning(r["g"]["text"][:1].cuda(), r["g"]["audio"][1:2].cuda(), r["g"]["audio"][:1].cuda(), r["g"]["mask"], lat0.shape[2])
    want_lat = eng.run(lat0, r["steps"])
    want_frames = r["vae"].decode_to_uint8_frames(want_lat)
    torch.cuda.synchronize()
    # what the C++ host computes from the bundle and three input files
    for name, tns in (("text", r["text"]), ("audio", r["audio"]), ("latents", lat0)):
        _bytes(tns).cpu().numpy().tofile(str(tmp_path / f"{name}.in"))
    b, c, f, h, w = r["shape"]
    prog = tmp_path / "program.txt"
    prog.write_text(
        f"load text {tmp_path}/text.in\nload audio {tmp_path}/audio.in\nload latents {tmp_path}/latents.in\n"
        "run set_conditioning\n"
        f"denoise {tmp_path}/steps.bin latents x t noise_pred 2 4.0 0.0 {b} {c} {f} {h * w}\n"
        f"save latents {tmp_path}/latents.out\nrun decode\nsave frames {tmp_path}/frames.out\n")
    out = subprocess.run([host, _lib.LIB_PATHS["bf16"], r["path"], str(prog)], capture_output=True, text=True, timeout=300)
    print(out.stdout[-600:], out.stderr[-600:])
    assert out.returncode == 0
    got_lat = torch.from_numpy(np.fromfile(str(tmp_path / "latents.out"), dtype=np.float32)).reshape(r["shape"])
    got_frames = torch.from_numpy(np.fromfile(str(tmp_path / "frames.out"), dtype=np.uint8)).reshape(want_frames.shape)
    assert torch.equal(got_lat, want_lat.cpu()), f"max |diff| {float((got_lat - want_lat.cpu()).abs().max()):.3e}"
    assert torch.equal(got_frames, want_frames.cpu())
    assert torch.equal(got_lat[:, :, 0], lat0[:, :, 0].cpu())                 # frame 0 pinned by the host's loop too


def test_cpp_host_second_clip_and_fp16_library(tmp_path, fp16_library):
    """The same bundle serves every clip of its geometry (the conditioning plan recomputes the per-clip caches from the
    text / audio regions), and a bundle recorded with the fp16 library replays through libavsd_hip_f16.so — and is refused by
    the bf16 one."""
    from asva_amd import _lib, build
    from asva_amd import precision as P

    r = _record(tmp_path, "ddim", steps=3)
    eng, lat0, g = r["eng"], r["lat0"], r["g"]
    dev = lat0.device
    gen = torch.Generator().manual_seed(9)
    audio2 = torch.randn(1, 229, g["audio"].shape[-1], generator=gen)
    lat2 = torch.randn(lat0.shape, generator=gen).to(dev)
    eng.set_conditioning(g["text"][:1].cuda(), audio2.cuda(), g["audio"][:1].cuda(), g["mask"], lat0.shape[2])
    want2 = eng.run(lat2, r["steps"])
    eng.set_conditioning(g["text"][:1].cuda(), g["audio"][1:2].cuda(), g["audio"][:1].cuda(), g["mask"], lat0.shape[2])
    want1 = eng.run(lat0, r["steps"])
    torch.cuda.synchronize()
    audio2_cfg = torch.cat([g["audio"][:1], audio2]).to(dev, P.ACT).contiguous()
    for name, tns in (("text", r["text"]), ("audio1", r["audio"]), ("audio2", audio2_cfg), ("lat1", lat0), ("lat2", lat2)):
        _bytes(tns).cpu().numpy().tofile(str(tmp_path / f"{name}.in"))
    b, c, f, h, w = r["shape"]
    den = f"denoise {tmp_path}/steps.bin latents x t noise_pred 2 4.0 0.0 {b} {c} {f} {h * w}\n"
    prog = tmp_path / "program.txt"
    prog.write_text(f"load text {tmp_path}/text.in\n"
                    f"load audio {tmp_path}/audio2.in\nload latents {tmp_path}/lat2.in\nrun set_conditioning\n" + den +
                    f"save latents {tmp_path}/lat2.out\n"
                    f"load audio {tmp_path}/audio1.in\nload latents {tmp_path}/lat1.in\nrun set_conditioning\n" + den +
                    f"save latents {tmp_path}/lat1.out\n")
    out = subprocess.run([build.PLAN_HOST, _lib.LIB_PATHS["fp16"], r["path"], str(prog)], capture_output=True, text=True, timeout=300)
    print(out.stdout[-400:], out.stderr[-400:])
    assert out.returncode == 0
    for name, want in (("lat2", want2), ("lat1", want1)):
        got = torch.from_numpy(np.fromfile(str(tmp_path / f"{name}.out"), dtype=np.float32)).reshape(r["shape"])
        assert torch.equal(got, want.cpu()), name
    bad = subprocess.run([build.PLAN_HOST, _lib.LIB_PATHS["bf16"], r["path"], str(prog)], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "other storage precision" in bad.stderr
