"""Writes a launch-plan bundle (include/avsd.h "launch plans", asva_amd/plan.py) for the AVSync15 geometry of BASELINE.json
configs[1]: SD1.5 UNet3D (random-init, seeded), one 12 x 256 x 256 clip, audio-only guidance (CFG batch 2), 50-step PLMS
table, SD1.5 VAE decode — and the program file that makes tools/plan_host.cpp run the whole clip without Python.

    python tools/export_plan.py --out /tmp/avsync15 [--steps 50] [--no-vae]
    asva_amd/plan_host asva_amd/libavsd_hip.so /tmp/avsync15/clip.plan /tmp/avsync15/program.txt

Also runs the same clip through the Python DenoiseEngine and stores its final latents / frames (expected.*.bin) so the two
hosts can be compared byte for byte (tools/export_plan.py --check /tmp/avsync15 after plan_host has run).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def _bytes(t):
    return t.detach().contiguous().reshape(-1).view(torch.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/avsync15")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--check", default=None, help="compare <dir>/latents.out, frames.out (written by plan_host) with expected.*.bin")
    a = ap.parse_args()
    if a.check:
        ok = True
        for name in ("latents", "frames"):
            got, want = os.path.join(a.check, name + ".out"), os.path.join(a.check, "expected." + name + ".bin")
            if os.path.exists(got) and os.path.exists(want):
                same = np.array_equal(np.fromfile(got, dtype=np.uint8), np.fromfile(want, dtype=np.uint8))
                print(f"{name}: plan_host output {'==' if same else '!='} Python engine output")
                ok &= same
        sys.exit(0 if ok else 1)

    import bench
    from asva_amd import plan, precision as P, unet as U
    from asva_amd.conditioning import audio_segment_mask
    from asva_amd.engine import DenoiseEngine
    from asva_amd.schedulers import PNDMScheduler

    dev = torch.device("cuda", 0)
    os.makedirs(a.out, exist_ok=True)
    unet = bench.build_unet(dev, 0, 1)
    g = torch.Generator().manual_seed(0)
    lat0 = torch.randn(1, 4, 12, 32, 32, generator=g).to(dev)
    text1, audio1, null1 = (torch.randn(1, n, 768, generator=g) for n in (77, 229, 229))
    text = torch.cat([text1, text1]).to(dev, P.ACT).contiguous()
    audio = torch.cat([null1, audio1]).to(dev, P.ACT).contiguous()
    mask = audio_segment_mask(12)
    vae = None
    if not a.no_vae:
        from asva_amd.vae import AutoencoderKL

        with torch.device(dev):
            vae = AutoencoderKL(**bench.SD15_VAE).eval()
    x, t, latents = lat0.clone(), torch.full((1,), 501.0, device=dev), lat0.clone()
    rec = plan.Recorder()
    rec.region("unet_weights", unet.pack().blob, plan.CONST)
    rec.region("frame_index", U.frame_index(12, dev), plan.CONST)
    rec.region("audio_key_index", U.key_index_for(mask, dev), plan.CONST)
    if vae is not None:
        rec.region("vae_weights", vae.pack().blob, plan.CONST)
    # which kind of "forward" plan this is (include/avsd.h): text is [t, t] here, so with AVSD_SHARE_PREFIX on (default) the
    # layers in front of the first audio cross-attention are recorded once for both guidance branches
    share_flag = torch.tensor([int(U._SHARE_PREFIX), 0, 0, 0], dtype=torch.int32, device=dev)
    rec.region("share_prefix", share_flag, plan.CONST)
    for name, tns in (("text", text), ("audio", audio), ("x", x), ("t", t), ("latents", latents)):
        rec.region(name, tns, plan.INPUT)
    with rec.record("set_conditioning"):
        unet.set_conditioning(text, audio, mask, 12)
    with rec.record("forward"):
        noise = unet.denoise_forward(x, t, rep=2)
    rec.region("noise_pred", noise, plan.OUTPUT)
    if vae is not None:
        with rec.record("decode"):
            frames = vae.decode_to_uint8_frames(latents)
        rec.region("frames", frames, plan.OUTPUT)
    b = rec.save(os.path.join(a.out, "clip.plan"))
    print("bundle:", b.n_calls, f"launches; {len(b.buffer_sizes())} buffers, {sum(b.buffer_sizes()) / 1e9:.2f} GB")
    b.close()
    eng = DenoiseEngine(unet, PNDMScheduler(), audio_guidance_scale=4.0)
    eng.prepare(lat0, a.steps)
    plan.export_steps(os.path.join(a.out, "steps.bin"), eng._ts.tolist(), eng._plans)
    for name, tns in (("text", text), ("audio", audio), ("latents", lat0)):
        _bytes(tns).cpu().numpy().tofile(os.path.join(a.out, name + ".in"))
    prog = [f"load text {a.out}/text.in", f"load audio {a.out}/audio.in", f"load latents {a.out}/latents.in", "run set_conditioning",
            f"denoise {a.out}/steps.bin latents x t noise_pred 2 4.0 0.0 1 4 12 1024", f"save latents {a.out}/latents.out"]
    if vae is not None:
        prog += ["run decode", f"save frames {a.out}/frames.out"]
    open(os.path.join(a.out, "program.txt"), "w").write("\n".join(prog) + "\n")
    # the Python host on the same clip
    eng.set_conditioning(text1.to(dev), audio1.to(dev), null1.to(dev), mask, 12)
    want = eng.run(lat0, a.steps)
    _bytes(want).cpu().numpy().tofile(os.path.join(a.out, "expected.latents.bin"))
    if vae is not None:
        _bytes(vae.decode_to_uint8_frames(want)).cpu().numpy().tofile(os.path.join(a.out, "expected.frames.bin"))
    torch.cuda.synchronize()
    print(f"wrote {a.out}: clip.plan, clip.plan.d/, steps.bin ({len(eng._plans)} UNet evaluations), program.txt, expected.*.bin")


if __name__ == "__main__":
    main()
