"""Host logic on CPU (-m "not gpu"): state_dict/config surface, weight packing, layer sequencing and the
conditioning cache of asva_amd.unet, checked against golden vectors produced by the REFERENCE UNet
(tests/golden, oracle/gen_golden.py) with the kernels replaced by their CPU contract emulation
(tests/emu_ops.py)."""
import json
import os

import pytest
import torch

from tests import emu_ops
from tests.helpers import bf16_round_state_dict, filled_unet, load_golden, load_shapes, rel_l2


@pytest.fixture()
def emulated(monkeypatch):
    import asva_amd.unet as unet_mod

    monkeypatch.setattr(unet_mod, "ops", emu_ops)
    return unet_mod


def test_state_dict_surface_matches_reference():
    from asva_amd.unet import AudioUNet3DConditionModel

    tiny = load_golden("unet_tiny_e2e.pt")["config"]
    m = AudioUNet3DConditionModel.from_config(tiny)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == load_shapes("unet_tiny_state_dict_shapes.json")
    assert len(m.config) == 42 and m.config.in_channels == 4 and m.config["sample_size"] == 8
    # temporal paths start at zero exactly like the reference (utils.py:31-32, transformer :267)
    sd = m.state_dict()
    assert all(float(v.abs().sum()) == 0 for k, v in sd.items() if "conv_temp" in k)
    assert all(float(v.abs().sum()) == 0 for k, v in sd.items() if k.endswith("attn_temp.to_out.0.weight"))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "unet_sd15_state_dict_shapes.json")),
                    reason="full-shape fixture absent")
def test_sd15_state_dict_surface_matches_reference():
    from asva_amd.unet import AudioUNet3DConditionModel

    with torch.device("meta"):
        m = AudioUNet3DConditionModel.from_config(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "unet_sd15_config.json"))))
    shapes = load_shapes("unet_sd15_state_dict_shapes.json")
    assert len(shapes) == 1106
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes
    assert sum(v.numel() for v in m.state_dict().values()) == 1169357496


def test_unsupported_config_raises():
    from asva_amd.unet import AudioUNet3DConditionModel

    with pytest.raises(NotImplementedError):
        AudioUNet3DConditionModel(use_linear_projection=True)
    with pytest.raises(NotImplementedError):
        AudioUNet3DConditionModel(block_out_channels=(32, 64, 64, 64))   # head dim 4 has no kernel


def test_forward_without_library_or_gpu_fails_loudly():
    g = load_golden("unet_tiny_e2e.pt")
    m = filled_unet(g["config"])
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        m(g["sample"], 981, g["text"], g["audio"], audio_attention_mask=g["mask"])


def test_orchestration_matches_reference_golden(emulated):
    """bf16-storage emulation of the full forward vs the reference's fp32 output.  Tolerance 3e-2: the
    reference's own bf16-vs-fp32 gap on this network is 1.4e-2."""
    g = load_golden("unet_tiny_e2e.pt")
    m = filled_unet(g["config"])
    B, Fr = g["sample"].shape[0], g["sample"].shape[2]
    text = g["text"][:, None].expand(B, Fr, *g["text"].shape[1:])
    audio = g["audio"][:, None].expand(B, Fr, *g["audio"].shape[1:])
    mask = g["mask"][None].expand(B, -1, -1)
    for t, ref in zip(g["timesteps"], g["out"]):
        out = m(g["sample"], t, text, audio, audio_attention_mask=mask).sample
        assert out.shape == ref.shape and out.dtype == torch.float32
        err = rel_l2(out, ref)
        assert err < 3e-2, err
    # cached-conditioning call (the denoising-loop form) gives the same answer
    out2 = m(g["sample"], g["timesteps"][-1]).sample
    assert torch.equal(out2, out)


def test_orchestration_vs_oracle_on_rounded_weights(emulated):
    """Same comparison against the oracle run on bf16-rounded weights: isolates activation rounding."""
    from oracle.unet_ref import unet_forward

    g = load_golden("unet_tiny_e2e.pt")
    m = filled_unet(g["config"])
    B, Fr = g["sample"].shape[0], g["sample"].shape[2]
    text = g["text"][:, None].expand(B, Fr, *g["text"].shape[1:])
    audio = g["audio"][:, None].expand(B, Fr, *g["audio"].shape[1:])
    mask = g["mask"][None].expand(B, -1, -1)
    ref = unet_forward(bf16_round_state_dict(m.state_dict()), dict(m.config), g["sample"].to(torch.bfloat16).float(),
                       981, text.to(torch.bfloat16).float(), audio.to(torch.bfloat16).float(), mask)
    out = m(g["sample"], 981, text, audio, audio_attention_mask=mask).sample
    err = rel_l2(out, ref)
    assert err < 2e-2, err


def test_per_frame_conditioning_and_batch_masks(emulated):
    """Genuinely per-frame text/audio and per-batch masks take the general (q_per_kv = 1) path."""
    from oracle.unet_ref import unet_forward

    g = load_golden("unet_tiny_e2e.pt")
    m = filled_unet(g["config"])
    B, Fr = g["sample"].shape[0], g["sample"].shape[2]
    gen = torch.Generator().manual_seed(5)
    text = torch.randn(B, Fr, 7, 64, generator=gen)
    audio = torch.randn(B, Fr, 229, 64, generator=gen)
    mask = g["mask"][None].expand(B, -1, -1).clone()
    mask[1] = mask[1].roll(1, 0)               # batch 1 sees the chunks in a different frame order
    ref = unet_forward(m.state_dict(), dict(m.config), g["sample"], 501, text, audio, mask)
    out = m(g["sample"], 501, text, audio, audio_attention_mask=mask).sample
    assert rel_l2(out, ref) < 3e-2


def test_shared_guidance_prefix_is_exact(emulated, monkeypatch):
    """Audio-only guidance feeds both branches the same latents, timestep and text (pipeline_audio_cond_animation.py:155):
    the layers before the first audio cross-attention are computed once and replicated.  Replication is data movement, so
    the result equals the per-branch computation exactly; branches with different text must not take the shortcut."""
    from asva_amd import unet as U
    from asva_amd.conditioning import audio_segment_mask

    g = load_golden("unet_tiny_e2e.pt")
    m = filled_unet(g["config"])
    Fr = g["sample"].shape[2]
    gen = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 4, Fr, *g["sample"].shape[3:], generator=gen)
    text1 = torch.randn(1, g["text"].shape[1], g["text"].shape[2], generator=gen)
    audio = torch.randn(2, Fr, 229, g["audio"].shape[-1], generator=gen)
    t = torch.full((1,), 501.0)
    calls = []
    real = U._replicate
    monkeypatch.setattr(U, "_replicate", lambda a, r: (calls.append(r), real(a, r))[1])

    def run(text, share):
        monkeypatch.setattr(U, "_SHARE_PREFIX", share)
        m.set_conditioning(text, audio, audio_segment_mask(Fr) if g["mask"] is None else g["mask"], Fr)
        return m.denoise_forward(lat, t, rep=2)

    same = torch.cat([text1, text1])
    a = run(same, True)
    assert calls and all(r == 2 for r in calls)
    n = len(calls)
    b = run(same, False)
    assert len(calls) == n and torch.equal(a, b)
    diff = torch.cat([text1, torch.zeros_like(text1)])
    c = run(diff, True)
    assert len(calls) == n, "branches with different text must run the prefix per branch"
    assert not torch.equal(a, c)


def test_save_and_from_pretrained_roundtrip(tmp_path, emulated):
    from asva_amd.unet import AudioUNet3DConditionModel

    g = load_golden("unet_tiny_e2e.pt")
    m = filled_unet(g["config"])
    m.save_pretrained(str(tmp_path / "modules" / "unet"))
    m2 = AudioUNet3DConditionModel.from_pretrained(str(tmp_path / "modules"), subfolder="unet")
    assert dict(m2.config) == dict(m.config)
    for (k1, v1), (k2, v2) in zip(sorted(m.state_dict().items()), sorted(m2.state_dict().items())):
        assert k1 == k2 and torch.equal(v1, v2)


# ---- VAE decoder host logic -----------------------------------------------------------------------------------
TINY_VAE = dict(block_out_channels=(32, 64, 128, 128), layers_per_block=2, norm_num_groups=32, latent_channels=4,
                in_channels=3, out_channels=3, scaling_factor=0.18215)


def _filled_vae(cfg):
    from asva_amd.vae import AutoencoderKL
    from oracle.filler import fill_module_

    m = AutoencoderKL.from_config(cfg).eval()
    fill_module_(m)
    return m


def test_vae_state_dict_surface_matches_sd15():
    from asva_amd.vae import AutoencoderKL
    from oracle.vae_ref import SD15_VAE_CONFIG, decoder_shapes, encoder_shapes

    with torch.device("meta"):
        m = AutoencoderKL.from_config(SD15_VAE_CONFIG)
    want = dict(decoder_shapes(SD15_VAE_CONFIG))
    want.update(encoder_shapes(SD15_VAE_CONFIG))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want
    assert sum(v.numel() for k, v in m.state_dict().items() if k.startswith(("decoder.", "post_quant"))) == 49490199
    assert sum(v.numel() for v in m.state_dict().values()) == 83653863          # the SD1.5 AutoencoderKL
    assert m.config.scaling_factor == 0.18215 and len(m.config.block_out_channels) == 4


def test_vae_encode_orchestration_vs_oracle(monkeypatch):
    import asva_amd.vae as vae_mod
    from oracle.vae_ref import vae_encode_moments

    monkeypatch.setattr(vae_mod, "ops", emu_ops)
    m = _filled_vae(TINY_VAE)
    x = torch.rand(2, 3, 32, 48, generator=torch.Generator().manual_seed(0)) * 2 - 1
    mean, logvar = vae_encode_moments(m.state_dict(), TINY_VAE, x)
    dist = m.encode(x).latent_dist
    assert dist.mean.shape == (2, 4, 4, 6)
    assert rel_l2(dist.mean, mean) < 3e-2 and rel_l2(dist.logvar, logvar) < 3e-2
    noise = torch.randn(2, 4, 4, 6, generator=torch.Generator().manual_seed(1))
    assert rel_l2(dist.sample(noise=noise), mean + torch.exp(0.5 * logvar) * noise) < 3e-2
    assert torch.equal(dist.mode(), dist.mean)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    assert torch.equal(dist.sample(generator=g1), dist.sample(generator=g2))


def test_vae_decode_orchestration_vs_oracle(monkeypatch):
    import asva_amd.vae as vae_mod
    from oracle.vae_ref import vae_decode

    monkeypatch.setattr(vae_mod, "ops", emu_ops)
    m = _filled_vae(TINY_VAE)
    z = torch.randn(3, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    ref = vae_decode(m.state_dict(), TINY_VAE, z)
    out = m.decode(z).sample
    assert out.shape == (3, 3, 64, 64)
    assert rel_l2(out, ref) < 3e-2
    # pipeline post-processing form and frame chunking
    lat = (z * 0.18215).reshape(1, 3, 4, 8, 8).permute(0, 2, 1, 3, 4).contiguous()
    vid = m.decode_to_video(lat)
    assert vid.shape == (1, 3, 3, 64, 64) and float(vid.min()) >= 0 and float(vid.max()) <= 1
    assert rel_l2(vid[0], (ref / 2 + 0.5).clamp(0, 1)) < 3e-2
    chunked = m.decode(z, frames_per_chunk=2).sample
    assert rel_l2(chunked, out) < 2e-2   # (different GEMM batch sizes: last-bit f32 differences flip bf16 roundings)


def test_vae_accepts_legacy_checkpoint_names():
    m = _filled_vae(TINY_VAE)
    legacy = {}
    for k, v in m.state_dict().items():
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f"attentions.0.{new}." in k:
                k = k.replace(f"attentions.0.{new}.", f"attentions.0.{old}.")
                if k.endswith("weight"):
                    v = v[:, :, None, None]
        legacy[k] = v
    assert any(".query." in k for k in legacy)
    m2 = _filled_vae(TINY_VAE)
    with torch.no_grad():
        for p in m2.parameters():
            p.zero_()
    m2.load_state_dict(legacy)
    for (k1, v1), (k2, v2) in zip(sorted(m.state_dict().items()), sorted(m2.state_dict().items())):
        assert k1 == k2 and torch.equal(v1, v2)


def test_from_pretrained_2d_inflation(tmp_path):
    """2-D -> 3-D inflation (audio_cond_unet_3d_condition.py:800-838): every key containing '_temp', every key missing
    from the 2-D checkpoint (audio attention) and every shape mismatch keeps the fresh 3-D init; the rest is loaded."""
    import json

    from safetensors.torch import save_file

    from asva_amd.unet import AudioUNet3DConditionModel

    g = load_golden("unet_tiny_e2e.pt")
    src = filled_unet(g["config"])
    sd2d = {k: v.clone() for k, v in src.state_dict().items() if "_temp" not in k and "audio" not in k}
    sd2d["conv_in.weight"] = torch.zeros(80, 9, 3, 3)                      # a shape mismatch must be ignored
    cfg2d = {k: (list(v) if isinstance(v, tuple) else v) for k, v in g["config"].items()}
    cfg2d.update(_class_name="UNet2DConditionModel", _diffusers_version="0.29.2",
                 down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], mid_block_type="UNetMidBlock2DCrossAttn",
                 up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3)
    cfg2d.pop("audio_cross_attention_dim")
    d = tmp_path / "sd" / "unet"
    d.mkdir(parents=True)
    json.dump(cfg2d, open(d / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in sd2d.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    cfg3d = dict(down_block_types=g["config"]["down_block_types"], up_block_types=g["config"]["up_block_types"],
                 mid_block_type=g["config"]["mid_block_type"], audio_cross_attention_dim=64)
    m = AudioUNet3DConditionModel.from_pretrained_2d(cfg3d, str(tmp_path / "sd"), subfolder="unet")
    assert m.config.down_block_types == tuple(g["config"]["down_block_types"]) and m.config.audio_cross_attention_dim == 64
    got = m.state_dict()
    for k, v in src.state_dict().items():
        if k == "conv_in.weight":
            assert not torch.equal(got[k], torch.zeros_like(got[k])) and got[k].shape == v.shape      # fresh init kept
        elif "conv_temp" in k or k.endswith("attn_temp.to_out.0.weight"):
            assert float(got[k].abs().sum()) == 0                                                     # zero-initialised temporal path
        elif "_temp" in k or "audio" in k:
            assert not torch.equal(got[k], v)                                                         # fresh, not the source's filler
        else:
            assert torch.equal(got[k], v), k


def test_non_audio_block_types_and_input_validation(emulated):
    """The reference also ships block types without the audio cross-attention (unet_3d_blocks.py:372-702,
    ff_spatio_temp_transformer_3d.py); they are constructible and run through the same kernels.  Also: the error
    behaviour of forward() on inputs the path does not take."""
    from asva_amd.unet import AudioUNet3DConditionModel
    from oracle.filler import fill_module_
    from oracle.unet_ref import unet_forward

    cfg = dict(load_golden("unet_tiny_e2e.pt")["config"])
    cfg["down_block_types"] = ["FFSpatioTempCrossAttnDownBlock3D"] * 3 + ["FFSpatioTempResDownBlock3D"]
    cfg["up_block_types"] = ["FFSpatioTempResUpBlock3D"] + ["FFSpatioTempCrossAttnUpBlock3D"] * 3
    cfg["mid_block_type"] = "FFSpatioTempCrossAttnUNetMidBlock3D"
    m = AudioUNet3DConditionModel.from_config(cfg).eval()
    fill_module_(m)
    assert not any("audio" in k for k in m.state_dict())
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 3, 8, 8, generator=gen)
    text = torch.randn(1, 7, 64, generator=gen)
    out = m(x, 77, text).sample                                        # 3-D text, no audio, no mask
    ref = unet_forward(m.state_dict(), cfg, x, 77, text[:, None].expand(1, 3, 7, 64), torch.zeros(1, 3, 229, 64), None)
    assert rel_l2(out, ref) < 3e-2
    # input validation
    g = load_golden("unet_tiny_e2e.pt")
    full = filled_unet(g["config"])
    with pytest.raises(ValueError, match="audio_encoder_hidden_states is required"):
        full(g["sample"], 1, g["text"])
    with pytest.raises(NotImplementedError, match="multiples of"):
        full(torch.zeros(2, 4, 4, 6, 8), 1, g["text"], g["audio"], audio_attention_mask=g["mask"])
    with pytest.raises(NotImplementedError, match="class_labels"):
        full(g["sample"], 1, g["text"], g["audio"], class_labels=torch.zeros(2))
    with pytest.raises(AssertionError):
        full(g["sample"][:, :, 0], 1, g["text"], g["audio"])
    full(g["sample"], 1, g["text"], g["audio"], audio_attention_mask=g["mask"])
    with pytest.raises(ValueError, match="conditioning was prepared for batch"):
        full(g["sample"][:1], 1)                                       # cached conditioning is for batch 2
    # an all-visible mask (or None) attends to all 229 keys
    a = full(g["sample"], 1, g["text"], g["audio"], audio_attention_mask=None).sample
    b = full(g["sample"], 1, g["text"], g["audio"], audio_attention_mask=torch.ones(4, 229, dtype=torch.bool)).sample
    assert torch.equal(a, b)
    # ragged masks (different number of visible keys per frame) cannot be a gather list
    ragged = g["mask"].clone()
    assert not bool(ragged[0, 6])
    ragged[0, 6] = True
    with pytest.raises(ValueError, match="different numbers of keys"):
        full(g["sample"], 1, g["text"], g["audio"], audio_attention_mask=ragged)


def test_scheduler_rejects_options_it_does_not_restate(tmp_path):
    """sample-changing options must not be swallowed (diffusers' DDIM default is clip_sample=True)."""
    import json

    from asva_amd.schedulers import DDIMScheduler, PNDMScheduler

    for bad in (dict(clip_sample=True), dict(thresholding=True), dict(trained_betas=[0.1, 0.2]), dict(rescale_betas_zero_snr=True),
                dict(some_new_option=3)):
        with pytest.raises(NotImplementedError):
            DDIMScheduler(**bad)
    # the SD1.5 scheduler_config.json (PNDM) loads
    cfg = {"_class_name": "PNDMScheduler", "_diffusers_version": "0.6.0", "beta_end": 0.012, "beta_schedule": "scaled_linear",
           "beta_start": 0.00085, "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1,
           "trained_betas": None, "clip_sample": False}
    (tmp_path / "scheduler").mkdir()
    (tmp_path / "scheduler" / "scheduler_config.json").write_text(json.dumps(cfg))
    s = PNDMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    s.set_timesteps(50)
    assert len(s.timesteps) == 51
    d = DDIMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    d.set_timesteps(25)
    assert len(d.timesteps) == 25


def test_pndm_step_textbook_mode_keeps_a_copy():
    """cur_sample_aliases_latents=False: `.step()` (object protocol) and `plan_step` (engine) agree — the sample restored
    at the repeated timestep is the one passed to step 0 even when the caller overwrites its storage in place."""
    from asva_amd.schedulers import PNDMScheduler

    for alias in (True, False):
        s = PNDMScheduler(cur_sample_aliases_latents=alias)
        s.set_timesteps(10)
        g = torch.Generator().manual_seed(0)
        x = torch.randn(1, 4, 3, 4, 4, generator=g)
        x0 = x.clone()
        e0, e1 = torch.randn(x.shape, generator=g), torch.randn(x.shape, generator=g)
        x[:] = s.step(e0, s.timesteps[0], x).prev_sample          # in-place write-back, like the reference loop (:364)
        x1 = x.clone()
        out = s.step(e1, s.timesteps[1], x).prev_sample
        p = s.plan_step(1)
        want = p.ca * (x1 if alias else x0) + p.cb * (0.5 * e1 + 0.5 * e0)
        assert torch.allclose(out, want, atol=1e-6)


def test_data_utils_image_audio_video_and_lists(tmp_path, monkeypatch):
    """avgen.data.utils loaders (reference avgen/data/utils.py:118-470): value ranges, shapes, clip sampling, lists."""
    import numpy as np
    from PIL import Image
    from scipy.io import wavfile

    from avgen.data.utils import (get_evaluation_data, load_audio_clips_uniformly, load_av_clips_uniformly, load_image,
                                  load_video_clips_uniformly)

    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (300, 400, 3), dtype=np.uint8)).save(tmp_path / "im.png")
    im = load_image(str(tmp_path / "im.png"), (256, 256))
    assert im.shape == (3, 256, 256) and 0.0 <= float(im.min()) and float(im.max()) <= 1.0
    # centre crop to the target aspect ratio first: a constant-column image stays constant per column
    grad = np.tile(np.arange(400, dtype=np.uint8)[None, :, None] // 2, (300, 1, 3))
    Image.fromarray(grad).save(tmp_path / "g.png")
    gi = load_image(str(tmp_path / "g.png"), (256, 256))
    assert float(gi.std(dim=1).max()) < 1e-6 and float(gi[0, 0, 0]) * 255 >= 24.0     # (400-300)//2 = 50 columns trimmed

    sr = 22050
    wave = (np.sin(np.arange(sr * 5) * 2 * np.pi * 440 / sr) * 0.5 * 32767).astype(np.int16)
    wavfile.write(tmp_path / "a.wav", sr, wave)
    clips = load_audio_clips_uniformly(str(tmp_path / "a.wav"), 2.0, 3, load_audio_as_melspectrogram=False)
    assert len(clips) == 3 and all(c.shape == (1, 32000) for c in clips)
    # (load_audio_as_melspectrogram=True runs avsd_kaldi_fbank on the device: covered by tests/test_audio_gpu.py)

    frames = rng.integers(0, 255, (60, 48, 64, 3), dtype=np.uint8)            # 2 s at 30 fps
    frames[:, 0, 0, 0] = np.arange(60)                                         # frame id in a pixel
    audio = rng.standard_normal((1, 32000)).astype(np.float32) * 0.1
    np.savez(tmp_path / "v.npz", frames=frames, fps=30.0, audio=audio, audio_sr=16000)
    vid, aud = load_av_clips_uniformly(str(tmp_path / "v.npz"), video_fps=6, video_num_frame=6, image_size=(48, 64), num_clips=2,
                                       load_audio_as_melspectrogram=False)
    assert vid.shape == (2, 6, 3, 48, 64) and len(aud) == 2 and aud[0].shape == (1, 16000)
    ids = (vid[0, :, 0, 0, 0] * 255).round().int().tolist()
    assert ids == [0, 5, 10, 15, 20, 25]                                       # first frame at or after each multiple of 1/6 s
    assert load_video_clips_uniformly(str(tmp_path / "v.npz"), 6, 6, (48, 64), 1).shape == (1, 6, 3, 48, 64)
    with pytest.raises(RuntimeError, match="torchvision"):
        load_av_clips_uniformly(str(tmp_path / "v.mp4"))

    root = tmp_path / "datasets" / "AVSync15"
    root.mkdir(parents=True)
    (root / "test.txt").write_text("dog/a.mp4\ncat/b.mp4\n")
    monkeypatch.setenv("AVSD_DATASETS_ROOT", str(tmp_path / "datasets"))
    vroot, paths, cats, kind = get_evaluation_data("AVSync15")
    assert vroot.endswith("AVSync15/videos") and paths == ["dog/a.mp4", "cat/b.mp4"] and cats == ["dog", "cat"] and kind == "video"


def test_mjpeg_avi_writer_roundtrip(tmp_path):
    from asva_amd.pipeline import write_video
    from asva_amd.video_io import read_mjpeg_avi

    g = torch.Generator().manual_seed(0)
    base = torch.rand(1, 16, 16, 3, generator=g)
    video = (torch.nn.functional.interpolate(base.permute(0, 3, 1, 2), size=(64, 96), mode="bilinear").permute(0, 2, 3, 1)
             .expand(5, -1, -1, -1) * 255).to(torch.uint8).contiguous()
    audio = torch.sin(torch.arange(16000) * 0.05)[None] * 0.3
    path = write_video(str(tmp_path / "clip.mp4"), video, 6, audio, 16000, "aac")
    assert path.endswith("clip.avi")
    v, fps, a, afps = read_mjpeg_avi(path)
    assert v.shape == video.shape and abs(fps - 6.0) < 1e-6 and afps == 16000 and a.shape == audio.shape
    assert (v.float() - video.float()).abs().mean() < 8.0 and (a - audio).abs().max() < 1e-4


def test_split_precision_twin_storage_and_packing():
    """Host convention of the split-precision mode (asva_amd/precision.py): a split tensor is a view into the first half of its
    storage, its rest plane sits at the same offset in the second half; weights.to_act builds such pairs, views keep their rest
    plane, and Packer.finish lays the whole blob out the same way.  (No kernel runs here: -m gpu covers the arithmetic.)"""
    from asva_amd import precision as P
    from asva_amd import unet as U
    from asva_amd import weights as W
    from tests import emu_ops

    w = torch.randn(24, 40, generator=torch.Generator().manual_seed(0))
    assert not W.is_twin(W.to_act(w))                     # mode off: plain 16-bit tensor
    P.set_split(True)
    old_ops = U.ops
    try:
        t = W.to_act(w)
        assert t.dtype == P.ACT and t.shape == w.shape and W.is_twin(t)
        main, rest = t.float(), W.rest_of(t).float()
        assert torch.equal(main, w.to(P.ACT).float()) and torch.equal(rest, (w - main).to(P.ACT).float())
        v = W.from_act(t)
        assert ((v - w).norm() / w.norm()).item() < 2.0 ** -16 and ((main - w).norm() / w.norm()).item() > 1e-3
        sl = t[4:9, 8:24]                                  # a view carries its rest plane
        assert W.is_twin(sl) and torch.equal(W.from_act(sl), v[4:9, 8:24])
        assert not W.is_twin(torch.zeros(8, 8, dtype=P.ACT))
        # the packed blob of a model is one twin allocation: rest planes at the same offsets in the second half
        U.ops = emu_ops
        g = load_golden("unet_tiny_e2e.pt")
        m = filled_unet(g["config"])
        pk = m.pack("cpu")
        assert pk.split and pk.blob.numel() % 2 == 0
        half = pk.blob.numel() // 2
        wq = pk.down[0].attentions[0].attn1.wq_ln
        assert W.is_twin(wq) and wq.untyped_storage().data_ptr() == pk.blob.untyped_storage().data_ptr()
        off = wq.storage_offset() * 2
        assert torch.equal(pk.blob[half + off:half + off + wq.numel() * 2].view(P.ACT).view(wq.shape), W.rest_of(wq))
        b = m.down_blocks[0].attentions[0].transformer_blocks[0]
        want = b.attn1.to_q.weight.detach().float() * b.norm1.weight.detach().float()[None, :]
        assert ((W.from_act(wq) - want).norm() / want.norm()).item() < 2.0 ** -16
        # leaving the mode repacks
        P.set_split(False)
        assert not m.pack("cpu").split
    finally:
        P.set_split(False)
        U.ops = old_ops


def test_product_filler_matches_the_reference_side_filler():
    """asva_amd/filler.py (what bench.py fills its parity model with) and oracle/filler.py (what filled the REFERENCE model when
    the goldens were generated) are two statements of one rule: identical bits for every kind of parameter."""
    from asva_amd import filler as prod
    from oracle import filler as ref

    for name, shape in (("conv_in.weight", (320, 4, 3, 3)), ("conv_in.bias", (320,)), ("down_blocks.0.resnets.0.norm1.weight", (320,)),
                        ("mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight", (10240, 1280)),
                        ("up_blocks.3.resnets.2.conv1.conv_temp.weight", (320, 960))):
        assert torch.equal(prod.fill_tensor(name, shape), ref.fill_tensor(name, shape)), name
    assert torch.equal(prod.seeded_randn(7, 3, 5), ref.seeded_randn(7, 3, 5))


def test_pack_frag_layout_matches_the_header():
    """include/avsd.h AVSD_GEMM_W_FRAG: element (f, s, l, e) = W[32 f + (l & 31)][16 s + 8 (l >> 5) + e] (what csrc/nstream.hip streams);
    the nstream rule admits exactly the GEGLU projections of the 320- / 640-channel levels"""
    from asva_amd.weights import pack_frag
    from asva_amd import ops
    N, K = 96, 64
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K).to(torch.bfloat16)       # (values repeat after rounding: compare indices instead)
    idx = torch.arange(N * K, dtype=torch.int32).reshape(N, K)
    pf = pack_frag(idx)
    assert pf.shape == (N // 32, K // 16, 64, 8) and pf.is_contiguous()
    for f, s, l, e in [(0, 0, 0, 0), (1, 2, 5, 3), (2, 3, 63, 7), (0, 1, 32, 0), (2, 0, 31, 4)]:
        assert pf[f, s, l, e].item() == idx[32 * f + (l & 31), 16 * s + 8 * (l >> 5) + e].item()
    assert pack_frag(w).dtype == torch.bfloat16
    assert ops.nstream_supported(24576, 2560, 320) and ops.nstream_supported(6144, 5120, 640) and ops.nstream_supported(96, 1280, 320)
    assert not ops.nstream_supported(1536, 10240, 1280) and not ops.nstream_supported(24576, 960, 320) and not ops.nstream_supported(24576, 320, 320)


def test_precision_plan_wiring_needs_no_split_launch(emulated):
    """Per-layer precision plan (asva_amd/precision.py) on the kernel-contract emulation: every (main, rest) pair a three-pass product reads
    comes out of the epilogue that produced the tensor (AVSD_GEMM_OUT_REST / the two-plane outputs of the three-pass products themselves) —
    no avsd_split_f32 pass over an f32 master (round 5 launched 68 of them per step) — and the result stays closer to the reference's fp32
    output than the same storage without the plan's three-pass products."""
    from asva_amd import precision as P

    g = load_golden("unet_tiny_e2e.pt")
    B, Fr = g["sample"].shape[0], g["sample"].shape[2]
    text = g["text"][:, None].expand(B, Fr, *g["text"].shape[1:])
    audio = g["audio"][:, None].expand(B, Fr, *g["audio"].shape[1:])
    mask = g["mask"][None].expand(B, -1, -1)
    errs = {}
    try:
        for name in ("plan", "none"):
            P.set_plan(True)
            if name == "none":
                P.PLAN = {"three_pass": frozenset(), "name": "none"}
            m = filled_unet(g["config"])
            emu_ops.COUNTS.update(three_pass=0, split_planes=0)
            out = m(g["sample"], 981, text, audio, audio_attention_mask=mask).sample
            errs[name] = rel_l2(out, g["out"][0])
            if name == "plan":
                assert emu_ops.COUNTS["split_planes"] == 0, emu_ops.COUNTS
                assert emu_ops.COUNTS["three_pass"] >= 20, emu_ops.COUNTS      # conv_in / conv_out / shortcuts / samplers, each with its temporal mix
            else:
                assert emu_ops.COUNTS["three_pass"] == 0
    finally:
        P.set_plan(False)
    assert errs["plan"] < 1.2e-3 and errs["plan"] < 0.8 * errs["none"], errs


@pytest.mark.parametrize("mode", ["bf16", "split", "plan"])
def test_meta_replica_packs_the_same_blob_layout(mode):
    """asva_amd.dist: non-zero ranks build the layout from meta parameters and receive the bytes by ONE broadcast — the item list (sizes, offsets)
    must not depend on whether a tensor holds data (round-5 advisor finding: an `is_twin` test skipped an item on the data-holding rank only)."""
    from asva_amd import precision as P
    from asva_amd import unet as U
    from asva_amd.unet import AudioUNet3DConditionModel

    cfg = dict(load_golden("unet_tiny_e2e.pt")["config"])
    cfg["block_out_channels"] = (320, 640, 640, 640)        # 320 / 640 channels: the fragment-ordered GEGLU copy is registered
    cfg["attention_head_dim"] = 8
    old_ops = U.ops
    try:
        U.ops = emu_ops
        if mode == "split":
            P.set_split(True)
        elif mode == "plan":
            P.set_plan(True)
        real = AudioUNet3DConditionModel.from_config(cfg).pack("cpu")
        with torch.device("meta"):
            meta_model = AudioUNet3DConditionModel.from_config(cfg)
        pr = U.Packer()
        pk = U._Pk(conv_in=pr.ffconv(meta_model.conv_in, "conv_in"), t1=pr.lin(meta_model.time_embedding.linear_1), t2=pr.lin(meta_model.time_embedding.linear_2),
                   down=[pr.block(b, f"down_blocks.{i}") for i, b in enumerate(meta_model.down_blocks)], mid=pr.block(meta_model.mid_block, "mid_block"),
                   up=[pr.block(b, f"up_blocks.{i}") for i, b in enumerate(meta_model.up_blocks)],
                   norm_out=pr.aff(meta_model.conv_norm_out), conv_out=pr.ffconv(meta_model.conv_out, "conv_out"))
        meta = pr.finish(pk, "cpu", meta=True)
        assert meta.blob.numel() == real.blob.numel()
        a = real.down[0].attentions[0]
        b = meta.down[0].attentions[0]
        assert hasattr(a, "w1_ln_f") == hasattr(b, "w1_ln_f") == (mode != "split")
        assert a.ff2.w.storage_offset() == b.ff2.w.storage_offset() and a.ff2.w.shape == b.ff2.w.shape
    finally:
        P.set_split(False)
        P.set_plan(False)
        U.ops = old_ops


@pytest.mark.parametrize("hs,ws", [(4, 4), (3, 5)])
def test_subpixel_upsample_conv_packing_is_the_same_function(hs, ws):
    """weights.subpixel_conv3x3: nearest-2x upsample + 3x3 pad-1 convolution (FFSpatioTempResUpsample3D, ff_spatio_temp_resnet_3d.py:48-55)
    folded into four per-parity 2x2 kernels on the original image — checked against F.interpolate + F.conv2d in f64 through the contract
    the kernel implements (tests/emu_ops.py states it: AVSD_GEMM_CONV3 with ups = 2)."""
    import torch.nn.functional as F
    from asva_amd import precision as P
    from asva_amd.weights import subpixel_conv3x3

    g = torch.Generator().manual_seed(5)
    n_img, cin, cout = 3, 64, 64
    x = torch.randn(n_img, cin, hs, ws, generator=g).to(P.ACT)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), w.double(), b.double(), padding=1)
    wp = subpixel_conv3x3(w.permute(0, 2, 3, 1).contiguous())
    assert wp.shape == (4 * cout, 4 * cin)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    out = emu_ops.gemm(rows, wp.to(P.ACT), bias=b.repeat(4), mode=emu_ops.CONV3, conv=(n_img, hs, ws, 1, 2), out_f32=True)
    got = out.reshape(n_img, 2 * hs, 2 * ws, cout).permute(0, 3, 1, 2)
    # the only difference is the rounding of the (summed) weights to 16 bits
    assert rel_l2(got, ref.float()) < 4e-3
    exact = emu_ops.gemm(rows, wp.to(P.ACT), bias=b.repeat(4), mode=emu_ops.CONV3, conv=(n_img, hs, ws, 1, 2), out_f32=True,
                         a_rest=torch.zeros_like(rows), w_rest=(wp - wp.to(P.ACT).float()).to(P.ACT))
    assert rel_l2(exact.reshape(n_img, 2 * hs, 2 * ws, cout).permute(0, 3, 1, 2), ref.float()) < 3e-5


def test_smoke_guard_is_tied_to_measured_values():
    """__graft_entry__.smoke() asserts the 16-bit modes at `guard` x what they measured on MI355X (tests/golden/smoke_measured.json): the
    bounds must stay tight (a fixed 3e-2 hid a 23 % drift of the bf16 path in round 5) and smoke() must read them from this file"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "smoke_measured.json")) as f:
        sm = json.load(f)
    assert 1.0 < sm["guard"] <= 1.3 and set(sm["measured"]) == {"bf16", "fp16_f32res"}
    assert sm["guard"] * sm["measured"]["bf16"] < 3e-2 and sm["guard"] * sm["measured"]["fp16_f32res"] < 3e-3
    src = open(os.path.join(root, "__graft_entry__.py")).read()
    assert "smoke_measured.json" in src and 'bound("bf16")' in src and 'bound("fp16_f32res")' in src
