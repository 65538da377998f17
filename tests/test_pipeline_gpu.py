"""VAE decode and the whole generation loop on the MI355X (-m gpu) against the oracle.  Tolerances as in
test_unet_gpu.py: bf16 storage / f32 accumulation vs the fp32 oracle, 3e-2 for one network evaluation, 5e-2 after a
handful of chained denoising steps."""
import pytest
import torch

from tests.helpers import filled_unet, load_golden, rel_l2
from tests.test_host_cpu import TINY_VAE, _filled_vae

pytestmark = pytest.mark.gpu


def test_sd15_vae_decoder_matches_oracle():
    from oracle.vae_ref import SD15_VAE_CONFIG, vae_decode

    vae = _filled_vae(SD15_VAE_CONFIG)
    z = torch.randn(3, 4, 16, 16, generator=torch.Generator().manual_seed(0))
    ref = vae_decode(vae.state_dict(), SD15_VAE_CONFIG, z)
    vae = vae.to("cuda")
    out = vae.decode(z.cuda()).sample
    assert out.shape == (3, 3, 128, 128) and out.dtype == torch.float32
    err = rel_l2(out, ref)
    print(f"SD1.5 VAE decode (3 x 16x16 latents): rel-L2 vs oracle {err:.3e}")
    assert err < 2e-2          # 1.5 x the measured 1.3e-2; the fp16 and split-precision decoders: tests/test_precise_gpu.py
    lat = (z * 0.18215).reshape(1, 3, 4, 16, 16).permute(0, 2, 1, 3, 4).contiguous().cuda()
    vid = vae.decode_to_video(lat)
    assert rel_l2(vid[0], (ref / 2 + 0.5).clamp(0, 1)) < 2e-2 and float(vid.min()) >= 0 and float(vid.max()) <= 1


@pytest.mark.parametrize("split", [False, True])
def test_sd15_vae_encoder_matches_oracle(split):
    """split = True: the encoder (stride-2 3x3 convolutions with the asymmetric (0, 1) padding, mid-block attention, 128 x 96 input) in
    split precision against the fp32 oracle at 1.5e-4."""
    from asva_amd import precision as P

    P.set_split(split)
    try:
        _vae_encoder_vs_oracle(1.5e-4 if split else 3e-2)
    finally:
        P.set_split(False)


def _vae_encoder_vs_oracle(tol):
    from oracle.vae_ref import SD15_VAE_CONFIG, vae_encode_moments

    vae = _filled_vae(SD15_VAE_CONFIG)
    x = torch.rand(2, 3, 128, 96, generator=torch.Generator().manual_seed(0)) * 2 - 1
    mean, logvar = vae_encode_moments(vae.state_dict(), SD15_VAE_CONFIG, x)
    vae = vae.to("cuda")
    dist = vae.encode(x.cuda()).latent_dist
    e1, e2 = rel_l2(dist.mean, mean), rel_l2(dist.logvar, logvar)
    print(f"SD1.5 VAE encode (2 x 128x96): mean rel-L2 {e1:.3e}, logvar rel-L2 {e2:.3e}")
    assert dist.mean.shape == (2, 4, 16, 12) and e1 < tol and e2 < tol
    # encode -> decode round trip stays finite and image-shaped
    out = vae.decode(dist.mode()).sample
    assert out.shape == (2, 3, 128, 96) and bool(torch.isfinite(out).all())


def test_vae_decode_full_clip_shape_is_finite_and_chunking_is_consistent():
    from oracle.vae_ref import SD15_VAE_CONFIG

    vae = _filled_vae(SD15_VAE_CONFIG).to("cuda")
    z = torch.randn(12, 4, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
    out = vae.decode(z).sample                                   # 12 x 256 x 256: the cfg-2 clip
    assert out.shape == (12, 3, 256, 256) and bool(torch.isfinite(out).all())
    part = vae.decode(z, frames_per_chunk=5).sample
    assert rel_l2(part, out) < 2e-2   # other chunk size -> other autotuned tiles -> other f32 summation order -> bf16 flips
    torch.cuda.synchronize()
    import time

    t0 = time.time()
    vae.decode(z)
    torch.cuda.synchronize()
    print(f"VAE decode 12x256x256: {(time.time() - t0) * 1e3:.1f} ms")


@pytest.mark.parametrize("kind,split", [("pndm", False), ("ddim", False), ("pndm", True), ("ddim", True)])
def test_pipeline_matches_oracle_pipeline(kind, split):
    """split = True: the same generation in split precision, held to 1.5e-4 (measured 5.4e-5 latents / 3.2e-5 frames after 4 steps;
    north_star's bar is 1e-3) — a defect of that size in the scheduler glue, the guidance mix, the latent preparation or the decoder
    cannot hide behind 16-bit rounding there."""
    from asva_amd import precision as P

    P.set_split(split)
    try:
        _pipeline_vs_oracle(kind, 1.5e-4 if split else 5e-2, 1e-4 if split else 4e-2)   # (engine vs eager loop: a regression guard, 1.5x what was measured)
    finally:
        P.set_split(False)


def _pipeline_vs_oracle(kind, tol, tol_eager):
    from asva_amd.pipeline import AudioCondAnimationPipeline
    from asva_amd.schedulers import DDIMScheduler, PNDMScheduler
    from oracle import pipeline_ref

    g = load_golden("unet_tiny_e2e.pt")
    f, h, w = g["sample"].shape[2:]
    gen = torch.Generator().manual_seed(0)
    il, noise = torch.randn(1, 4, h, w, generator=gen) * 0.18215, torch.randn(1, 4, f - 1, h, w, generator=gen)
    unet, vae = filled_unet(g["config"]), _filled_vae(TINY_VAE)
    unet_sd = {k: v.clone() for k, v in unet.state_dict().items()}        # CPU copies for the oracle (pipe.to moves the modules)
    steps = 4
    x0 = pipeline_ref.prepare_video_latents(il, noise)
    ref_lat = pipeline_ref.denoise(unet_sd, dict(unet.config), x0, g["text"][:1], g["audio"][1:2], g["audio"][:1], g["mask"],
                                   steps, 4.0, kind)
    ref_vid = pipeline_ref.decode({k: v.clone() for k, v in vae.state_dict().items()}, TINY_VAE, ref_lat)
    pipe = AudioCondAnimationPipeline(unet=unet, scheduler=PNDMScheduler() if kind == "pndm" else DDIMScheduler(), vae=vae)
    pipe.to("cuda")
    pipe.set_progress_bar_config(disable=True)
    kw = dict(texts=[""], text_encodings=[g["text"][:1]], video_length=f, height=h * 8, width=w * 8, num_inference_steps=steps,
              audio_guidance_scale=4.0, image_latents=il, audio_encodings=g["audio"][1:2], null_audio_encodings=g["audio"][:1],
              audio_masks=g["mask"], noise=noise)
    lat = pipe(**kw, output_latents=True)
    assert torch.equal(lat[:, :, 0].cpu(), x0[:, :, 0])
    e1 = rel_l2(lat, ref_lat)
    vid = pipe(**kw)["videos"]
    e2 = rel_l2(vid, ref_vid)
    print(f"{kind}: latents after {steps} steps rel-L2 {e1:.3e}; decoded video rel-L2 {e2:.3e}")
    assert e1 < tol and e2 < tol
    assert vid.device.type == "cpu" and vid.shape == (1, f, 3, h * 8, w * 8)
    # tile selection is deterministic (committed table / static rule): the same call again is bit-identical
    assert torch.equal(pipe(**kw, output_latents=True), lat)
    # graph-replayed engine vs eager reference-style loop on the same UNet kernels.  Root cause of the 2-3e-2 (tools/engine_eager_ab.py,
    # round 5): the ROTATED K WALK.  The engine runs the still-identical guidance branches ONCE up to the first audio cross-attention (shared
    # prefix, M halves), the eager loop runs the full CFG batch; a row band's rotation start depends on the launch's row-tile count, so the
    # same rows are summed in another f32 order and a fraction of the 16-bit outputs rounds the other way — which four evaluations of this
    # tiny random network amplify to the size of its bf16 error itself.  With AVSD_KROT=0 the two paths agree to 8e-8 (6 seeds x 2
    # schedulers); the norm_temp fold changes nothing.  Not a defect: both are valid roundings; the split-precision rows below hold 1e-4.
    pipe.use_engine = False
    lat2 = pipe(**kw, output_latents=True)
    assert rel_l2(lat2, lat) < tol_eager
    # a second clip of the same geometry reuses the captured graph (conditioning refreshed in place)
    pipe.use_engine = True
    kw2 = dict(kw, audio_encodings=g["audio"][:1], null_audio_encodings=g["audio"][1:2])
    lat3 = pipe(**kw2, output_latents=True)
    ref3 = pipeline_ref.denoise(unet_sd, dict(unet.config), x0, g["text"][:1], g["audio"][:1], g["audio"][1:2], g["mask"],
                                steps, 4.0, kind)
    assert rel_l2(lat3, ref3) < tol


def test_generate_videos_batches_the_clips_of_a_video():
    """generate_videos(clips_per_forward=k): k clips per denoising run instead of the reference's one-by-one loop (:431-458).  Each
    clip starts from the noise the seed gives a single-clip call, so the frames equal the sequential ones up to the 16-bit
    rounding that another batch size (other tiles, other f32 summation order) re-draws."""
    from asva_amd.pipeline import AudioCondAnimationPipeline, generate_videos
    from asva_amd.schedulers import PNDMScheduler

    g = load_golden("unet_tiny_e2e.pt")
    f, h, w = g["sample"].shape[2:]
    pipe = AudioCondAnimationPipeline(unet=filled_unet(g["config"]), scheduler=PNDMScheduler(), vae=_filled_vae(TINY_VAE))
    pipe.to("cuda")
    pipe.set_progress_bar_config(disable=True)
    pipe.generation_steps = 4
    gen = torch.Generator().manual_seed(2)
    clips = [dict(image_latents=torch.randn(4, h, w, generator=gen) * 0.18215, audio_encodings=torch.randn(229, g["audio"].shape[-1], generator=gen),
                  null_audio_encodings=g["audio"][0]) for _ in range(3)]
    kw = dict(category="x", category_text_encoding=g["text"][:1], image_size=(h * 8, w * 8), video_num_frame=f, seed=5,
              device=torch.device("cuda"), clips=clips)
    seq, _ = generate_videos(pipe, **kw)
    bat, _ = generate_videos(pipe, **kw, clips_per_forward=3)
    assert len(seq) == len(bat) == 3 and all(v.dtype == torch.uint8 and v.shape == (f, h * 8, w * 8, 3) for v in bat)
    d = [float((a.float() - b.float()).abs().mean()) for a, b in zip(seq, bat)]
    between = float((seq[0].float() - seq[1].float()).abs().mean())
    print(f"batched vs sequential frames: mean |diff| {d} of 255; between two different clips {between:.1f}")
    assert max(d) < 3.0 and between > 4 * max(d)
