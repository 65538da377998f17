"""AudioCondAnimationPipeline and its drivers — MI355X-native mirror of
avgen/pipelines/pipeline_audio_cond_animation.py (class :30-375, generate_videos :378-468,
generate_videos_for_dataset :471-551): same constructor, `__call__` signature and return value, same driver
function names / keyword arguments, so `scripts/animation_gen.py` binds to it unchanged through the `avgen`
shim package at the repository root.

What runs where
  * denoising loop + guidance + scheduler + VAE decode: the gfx950 kernels (asva_amd.engine / unet / vae).
  * text / audio encoders (CLIP, ImageBind) and the VAE *encoder*: external, once-per-clip, out of this round's
    scope.  They are used through the reference's own object protocol when the
    caller supplies them; every one of them can be bypassed with precomputed tensors (`text_encodings`,
    `audio_encodings`, `null_audio_encodings`, `audio_masks`, `image_latents`, `noise`) — which is also how the
    parity tests inject identical latents/noise (the reference's RNG use is not reproducible across devices,
    SURVEY.md §4).
  * video / audio file I/O (torchvision, torchaudio — absent here): behind `load_*` / `write_video` hooks that
    raise a clear error when their backend is missing; `clips=` lets callers hand over decoded clips directly.
"""
from __future__ import annotations

import json
import os
from typing import Callable, List, Optional, Sequence, Tuple, Union

import torch

from . import dist as adist
from .conditioning import AUDIO_TOKENS, audio_segment_mask
from .engine import DenoiseEngine
from .schedulers import DDIMScheduler, PNDMScheduler
from .unet import AudioUNet3DConditionModel
from .vae import AutoencoderKL
from .audio_features import AudioMelspectrogramExtractor

MELSPECTROGRAM_SHAPE = (128, 204)   # pipeline_audio_cond_animation.py:77


class AudioCondAnimationPipeline:
    def __init__(self, text_encoder=None, tokenizer=None, unet: AudioUNet3DConditionModel = None, scheduler=None,
                 vae: AutoencoderKL = None, audio_encoder=None, null_text_encodings_path: str = ""):
        self.register_modules(text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, scheduler=scheduler, vae=vae,
                              audio_encoder=audio_encoder)
        if null_text_encodings_path:
            self.null_text_encoding = torch.load(null_text_encodings_path).view(1, 77, 768)
        self.melspectrogram_shape = MELSPECTROGRAM_SHAPE
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1) if vae is not None else 8
        self.audio_processor = AudioMelspectrogramExtractor()   # (:81) waveforms -> (b, 1, 128, 204) log-mel, on the device
        self._progress = {}
        self._device = torch.device("cpu")
        self._dtype = torch.float32
        self.use_engine = True               # fused hipGraph loop; False = reference-style Python loop

    # -- DiffusionPipeline-ish surface ---------------------------------------------------------------------------
    def register_modules(self, **mods):
        for k, v in mods.items():
            setattr(self, k, v)

    def to(self, torch_device=None, dtype=None, **kw):
        torch_device = torch_device if torch_device is not None else kw.get("device")
        for m in (self.text_encoder, self.unet, self.vae, self.audio_encoder):
            if m is not None and hasattr(m, "to"):
                if torch_device is not None:
                    m.to(torch_device)
                if dtype is not None and not isinstance(m, (AudioUNet3DConditionModel, AutoencoderKL)):
                    m.to(dtype)
        if torch_device is not None:
            self._device = torch.device(torch_device)
        if dtype is not None:
            self._dtype = dtype       # the HIP path computes in bf16/f32 regardless; dtype is the I/O dtype
        return self

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    def progress_bar(self, iterable):
        if self._progress.get("disable", False):
            return iterable
        try:
            from tqdm import tqdm

            return tqdm(iterable, **{k: v for k, v in self._progress.items() if k != "disable"})
        except Exception:
            return iterable

    # -- conditioning (reference :83-203) ---------------------------------------------------------------------------
    @torch.no_grad()
    def encode_text(self, texts, device, dtype, do_text_classifier_free_guidance, do_audio_classifier_free_guidance,
                    text_encodings=None):
        if text_encodings is None:
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("pass text_encodings=... or construct the pipeline with a tokenizer and text_encoder")
            ti = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                                return_tensors="pt")
            text_encodings = self.text_encoder(ti.input_ids.to(device))[0]
        elif isinstance(text_encodings, (list, tuple)):
            text_encodings = torch.cat(list(text_encodings))
        text_encodings = text_encodings.to(dtype=dtype, device=device)
        b = len(text_encodings)
        if do_text_classifier_free_guidance:
            if hasattr(self, "null_text_encoding"):
                uncond = self.null_text_encoding
            else:
                ui = self.tokenizer("", padding="max_length", max_length=text_encodings.shape[1], truncation=True, return_tensors="pt")
                uncond = self.text_encoder(ui.input_ids.to(device))[0]
            uncond = uncond.expand(b, -1, -1).to(dtype=dtype, device=device)
        if do_text_classifier_free_guidance and do_audio_classifier_free_guidance:
            return torch.cat([uncond, text_encodings, text_encodings])
        if do_text_classifier_free_guidance:
            return torch.cat([uncond, text_encodings])
        if do_audio_classifier_free_guidance:
            return torch.cat([text_encodings, text_encodings])
        return text_encodings

    @torch.no_grad()
    def encode_audio(self, audios, video_length=12, do_text_classifier_free_guidance=False,
                     do_audio_classifier_free_guidance=False, device=None, dtype=torch.float32, *,
                     audio_encodings=None, null_audio_encodings=None, audio_masks=None):
        """-> (encodings (k*b, 229, D) [NOT repeated over frames: the kernels share K/V across frames],
               masks (video_length, 229) bool).  Branch order as the reference (:186-194)."""
        if audio_encodings is None:
            if self.audio_encoder is None or self.audio_processor is None:
                raise ValueError("pass audio_encodings=/null_audio_encodings= or attach audio_encoder and audio_processor")
            mel = self.audio_processor(audios, device=device).to(dtype=dtype)
            _, audio_encodings, audio_masks = self.audio_encoder(mel, normalize=False, return_dict=False)
            if do_audio_classifier_free_guidance:
                null_mel = torch.zeros(1, 1, *self.melspectrogram_shape, device=device, dtype=dtype)
                _, null_audio_encodings, _ = self.audio_encoder(null_mel, normalize=False, return_dict=False)
        audio_encodings = audio_encodings.to(device=device, dtype=dtype)
        b = audio_encodings.shape[0]
        if audio_masks is None:
            audio_masks = audio_segment_mask(video_length)
        if audio_masks.dim() == 3:
            audio_masks = audio_masks[0]          # identical for every clip (segmask_imagebind.py:114)
        if do_audio_classifier_free_guidance:
            if null_audio_encodings is None:
                raise ValueError("audio guidance needs null_audio_encodings (encoding of the all-zero mel-spectrogram)")
            null = null_audio_encodings.to(device=device, dtype=dtype).expand(b, -1, -1)
        if do_text_classifier_free_guidance and do_audio_classifier_free_guidance:
            audio_encodings = torch.cat([null, null, audio_encodings])
        elif do_text_classifier_free_guidance:
            audio_encodings = torch.cat([audio_encodings, audio_encodings])
        elif do_audio_classifier_free_guidance:
            audio_encodings = torch.cat([null, audio_encodings])
        return audio_encodings, audio_masks

    @torch.no_grad()
    def encode_latents(self, image: torch.Tensor):
        return self.vae.encode(image.to(self.device)).latent_dist.sample() * self.vae.config.scaling_factor

    @torch.no_grad()
    def decode_latents(self, latents):
        """(b*f, 4, h, w) scaled latents -> (b*f, 3, H, W) f32 on the CPU in [0, 1] (reference :206-213)."""
        z = latents.to(torch.float32) / self.vae.config.scaling_factor
        return self.vae.decode(z, postprocess=True).sample.cpu().float()

    def prepare_video_latents(self, image_latents, num_channels_latents, video_length=12, height=256, width=256,
                              device=None, dtype=torch.float32, generator=None, noise=None):
        b = len(image_latents)
        shape = (b, num_channels_latents, video_length - 1, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if noise is None:
            noise = torch.randn(shape, generator=generator, device=device, dtype=dtype)
        noise = noise.to(device=device, dtype=dtype)
        assert tuple(noise.shape) == shape, (noise.shape, shape)
        lat = torch.cat([image_latents.unsqueeze(2).to(device=device, dtype=dtype), noise], dim=2)
        return lat * self.scheduler.init_noise_sigma

    # -- __call__ (reference :263-375) ---------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, images=None, audios=None, texts=None, text_encodings=None, video_length: int = 12, height: int = 256,
                 width: int = 256, num_inference_steps: int = 20, audio_guidance_scale: float = 4.0,
                 text_guidance_scale: float = 1.0, generator=None, return_dict: bool = True, *, image_latents=None,
                 audio_encodings=None, null_audio_encodings=None, audio_masks=None, noise=None, output_latents: bool = False,
                 output_type: str = "float"):
        device = self.device
        f32 = torch.float32
        do_text = text_guidance_scale > 1.0
        do_audio = audio_guidance_scale > 1.0
        k = 1 + int(do_text) + int(do_audio)

        text = self.encode_text(texts, device, f32, do_text, do_audio, text_encodings)                    # (k*b, 77, D)
        audio, masks = self.encode_audio(audios, video_length, do_text, do_audio, device, f32, audio_encodings=audio_encodings,
                                         null_audio_encodings=null_audio_encodings, audio_masks=audio_masks)
        if image_latents is None:
            if images is None:
                raise ValueError("pass images= (needs a VAE encoder) or image_latents=")
            image_latents = self.encode_latents(self._preprocess_images(images, height, width))
        image_latents = image_latents.to(device=device, dtype=f32)
        latents = self.prepare_video_latents(image_latents, self.unet.config.in_channels, video_length, height, width, device,
                                             f32, generator, noise)                                      # (b, 4, f, h, w)

        if self.use_engine and isinstance(self.scheduler, (PNDMScheduler, DDIMScheduler)):
            ekey = (id(self.unet), id(self.scheduler), float(audio_guidance_scale), float(text_guidance_scale))
            if getattr(self, "_engine_key", None) != ekey:      # keep the engine (and its captured graph) across clips
                self._engine = DenoiseEngine(self.unet, self.scheduler, audio_guidance_scale, text_guidance_scale)
                self._engine_key = ekey
            eng = self._engine
            self.unet.set_conditioning(text, audio, masks, video_length)      # already in the CFG branch order of :150-155,:186-194
            eng.prepare(latents, num_inference_steps)
            latents = latents.contiguous().clone()
            for i in self.progress_bar(range(self.scheduler.num_forwards())):
                eng.step(latents, i)
        else:
            latents = self._reference_style_loop(latents, text, audio, masks, video_length, num_inference_steps, do_text,
                                                 do_audio, audio_guidance_scale, text_guidance_scale)
        if output_latents:
            return latents
        if output_type == "uint8":          # (b, f, H, W, 3) uint8 CPU frames, converted on the device
            frames = self.vae.decode_to_uint8_frames(latents).cpu()
            return {"videos": frames} if return_dict else frames
        b, c, f, h, w = latents.shape
        videos = self.decode_latents(latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w))
        videos = videos.reshape(b, f, *videos.shape[1:])
        return {"videos": videos} if return_dict else videos

    def _reference_style_loop(self, latents, text, audio, masks, video_length, steps, do_text, do_audio, ag, tg):
        """The loop exactly as the reference writes it (:325-365): torch.cat duplication, unet(...).sample,
        guidance in torch, scheduler.step on frames 1.. — any scheduler object, any guidance mix."""
        self.scheduler.set_timesteps(steps, device=latents.device)
        k = 1 + int(do_text) + int(do_audio)
        txt = text[:, None].expand(-1, video_length, -1, -1)
        aud = audio[:, None].expand(-1, video_length, -1, -1)
        m = masks[None].expand(text.shape[0], -1, -1)
        latents = latents.clone()
        for t in self.progress_bar(self.scheduler.timesteps):
            x = torch.cat([latents] * k)
            x = self.scheduler.scale_model_input(x, t)
            n = self.unet(x, t, encoder_hidden_states=txt, audio_encoder_hidden_states=aud, audio_attention_mask=m).sample
            if do_text and do_audio:
                nu, nt, nta = n.chunk(3)
                n = nu + tg * (nt - nu) + ag * (nta - nt)
            elif do_text:
                na, nta = n.chunk(2)
                n = na + tg * (nta - na)
            elif do_audio:
                nt, nta = n.chunk(2)
                n = nt + ag * (nta - nt)
            latents[:, :, 1:] = self.scheduler.step(n[:, :, 1:], t, latents[:, :, 1:]).prev_sample
        return latents

    @staticmethod
    def _preprocess_images(images, height, width):
        """VaeImageProcessor.preprocess for PIL / (3,H,W) [0,1] tensors: -> (b, 3, H, W) in [-1, 1]."""
        out = []
        for im in images:
            if not torch.is_tensor(im):
                import numpy as np

                im = torch.from_numpy(np.asarray(im.convert("RGB").resize((width, height)))).permute(2, 0, 1).float() / 255.0
            out.append(im * 2.0 - 1.0)
        return torch.stack(out)


# ---- drivers ---------------------------------------------------------------------------------------------------------
from .data_utils import get_evaluation_data, load_audio_clips_uniformly, load_av_clips_uniformly, load_image  # noqa: E402


def write_video(filename, video_array, fps, audio_array=None, audio_fps=16000, audio_codec="aac"):
    """torchvision.io.write_video when torchvision exists (the reference's writer, :451-458: H.264 + AAC in .mp4);
    otherwise the codec-free Motion-JPEG / PCM writer of asva_amd.video_io, which keeps the requested file name's stem
    and writes `<stem>.avi` (this image has no H.264 encoder)."""
    try:
        import torchvision.io as tvio  # type: ignore
    except Exception as e:             # absent, or present but unusable (operator / ABI mismatch raises RuntimeError)
        import warnings

        from .video_io import write_mjpeg_avi

        path = os.path.splitext(filename)[0] + ".avi"
        warnings.warn(f"torchvision.io is unavailable ({type(e).__name__}): writing Motion-JPEG {path} instead of {filename}; "
                      "tools that glob *.mp4 must read it with asva_amd.video_io.read_mjpeg_avi", RuntimeWarning, stacklevel=2)
        return write_mjpeg_avi(path, video_array, fps, audio_array, audio_fps)
    tvio.write_video(filename=filename, video_array=video_array, fps=fps, audio_array=audio_array, audio_fps=audio_fps,
                     audio_codec=audio_codec)
    return filename


@torch.no_grad()
def generate_videos(pipeline, image_path: str = "", audio_path: str = "", video_path: str = "", category: str = "",
                    category_text_encoding: Optional[torch.Tensor] = None, image_size: Tuple[int, int] = (256, 256),
                    video_fps: int = 6, video_num_frame: int = 12, num_clips_per_video: int = 3,
                    audio_guidance_scale: float = 4.0, text_guidance_scale: float = 1.0, seed: int = 0, save_template: str = "",
                    device: torch.device = torch.device("cuda"), *, clips: Optional[Sequence[dict]] = None,
                    writer: Optional[Callable] = None, clips_per_forward: Optional[int] = None):
    """Reference generate_videos (:378-468).  `clips` (new, optional) = already-decoded inputs, one dict per clip with
    any of: image (3,H,W in [0,1]) or image_latents (4,h,w); audio (waveform) or audio_encodings (229,D) +
    null_audio_encodings (229,D)."""
    assert not (image_path and audio_path and video_path), "Can not specify image_path, audio_path, video_path all three"
    if clips is None:
        images = audios = None
        if image_path:
            images = [load_image(image_path, image_size)] * num_clips_per_video
        if audio_path:
            audios = load_audio_clips_uniformly(audio_path, video_num_frame / video_fps, num_clips_per_video,
                                                load_audio_as_melspectrogram=False)
        if video_path:
            vids, auds = load_av_clips_uniformly(video_path, video_fps, video_num_frame, image_size, num_clips_per_video,
                                                 load_audio_as_melspectrogram=False)
            images = images if images is not None else [v[0] for v in vids]
            audios = audios if audios is not None else auds
        clips = [{"image": im, "audio": au} for im, au in zip(images, audios)]
    videos, audios_out = [], []
    generator = torch.Generator(device=device)
    # clips_per_forward > 1 (AVSD_CLIPS_PER_FORWARD, default 1 = the reference's loop): that many clips of the video go through
    # ONE batched denoising run — 115 instead of 77 clip-steps/s on an MI355X (bench.py "batched").  Every clip still starts from
    # the noise the seed gives a single-clip call (:433 re-seeds per clip).  What differs from the one-clip-at-a-time loop: the
    # kernels' tile choice, and — when the image latents come from `images` rather than `image_latents` — the VAE posterior sample:
    # latent_dist.sample() draws for all clips of the group in one call from the global RNG, so clip k > 0 sees other draws.
    group = max(1, int(clips_per_forward if clips_per_forward is not None else os.environ.get("AVSD_CLIPS_PER_FORWARD", "1")))
    for k0 in range(0, len(clips), group):
        chunk = clips[k0:k0 + group]
        generator.manual_seed(seed)                       # every clip restarts from the same seed (:433)
        kw = {}
        for key in ("image_latents", "image", "audio_encodings"):      # a batched group must be homogeneous
            have = [key in c for c in chunk]
            if any(have) and not all(have):
                raise ValueError(f"generate_videos: clips {k0}..{k0 + len(chunk) - 1} are batched into one forward "
                                 f"(clips_per_forward={group}) but only some of them carry '{key}'")
        if all("image_latents" in c for c in chunk):
            kw["image_latents"] = torch.stack([c["image_latents"] for c in chunk])
        if all("audio_encodings" in c for c in chunk):
            kw["audio_encodings"] = torch.stack([c["audio_encodings"] for c in chunk])
            kw["null_audio_encodings"] = torch.stack([c["null_audio_encodings"] for c in chunk])
        if len(chunk) > 1:                                # the draw of a single-clip call (prepare_video_latents), shared by the group
            vsf = pipeline.vae_scale_factor
            one = torch.randn((1, pipeline.unet.config.in_channels, video_num_frame - 1, image_size[0] // vsf, image_size[1] // vsf),
                              generator=generator, device=device, dtype=torch.float32)
            kw["noise"] = one.expand(len(chunk), -1, -1, -1, -1)
        # uint8 (f, H, W, 3) frames = (video.permute(0, 2, 3, 1) * 255).byte() of the reference (:448), made on the device
        out = pipeline(images=[c["image"] for c in chunk] if all("image" in c for c in chunk) else None,
                       audios=[c.get("audio") for c in chunk], texts=[category] * len(chunk),
                       text_encodings=[category_text_encoding] * len(chunk) if category_text_encoding is not None else None,
                       video_length=video_num_frame, height=image_size[0], width=image_size[1],
                       num_inference_steps=getattr(pipeline, "generation_steps", 50),     # the reference hard-codes 50 (:442)
                       audio_guidance_scale=audio_guidance_scale, text_guidance_scale=text_guidance_scale,
                       generator=generator, return_dict=False, output_type="uint8", **kw)
        for j, clip in enumerate(chunk):
            video, k = out[j], k0 + j
            if save_template:
                path = f"{save_template}_clip-{k:02d}.mp4"
                os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
                (writer or write_video)(path, video, video_fps, clip.get("audio"), 16000, "aac")
            else:
                videos.append(video)
                audios_out.append(clip.get("audio"))
    if save_template:
        return None
    return videos, audios_out


@torch.no_grad()
def generate_videos_for_dataset(exp_root: str, checkpoint: int, dataset: str = "AVSync15", image_size: Tuple[int, int] = (256, 256),
                                video_fps: int = 6, video_num_frame: int = 12, num_clips_per_video: int = 3,
                                audio_guidance_scale: float = 4.0, text_guidance_scale: float = 1.0, random_seed: int = 0,
                                device: torch.device = torch.device("cuda"), dtype: torch.dtype = torch.float16):
    """Reference generate_videos_for_dataset (:471-551) with one addition: when launched with one process per GPU
    (torchrun) the video list is sharded by rank (clip i -> rank i mod world) — the reference loops sequentially."""
    from transformers import CLIPTextModel, CLIPTokenizer

    # one process per GPU (torchrun): every rank owns cuda:LOCAL_RANK.  The unchanged reference script passes
    # torch.device("cuda") = cuda:0, which would stack all ranks on one GPU.
    rank, local_rank, world = adist.env_rank_world()
    if world > 1 and torch.cuda.is_available() and torch.device(device).type == "cuda":
        idx = adist.device_index(local_rank)      # cuda:LOCAL_RANK; refuses more ranks than GPUs unless AVSD_DIST_SAME_DEVICE=1
        torch.cuda.set_device(idx)
        device = torch.device("cuda", idx)
    ckpt = f"{exp_root}/ckpts/checkpoint-{checkpoint}/modules"
    save_root = (f"{exp_root}/evaluations/checkpoint-{checkpoint}/AG-{audio_guidance_scale}_TG-{text_guidance_scale}/"
                 f"seed-{random_seed}/videos")
    video_root, filenames, categories, _ = get_evaluation_data(dataset)
    null_text = "./pretrained/openai-clip-l_null_text_encoding.pt"
    if dataset == "TheGreatestHits":
        enc = torch.load("./datasets/TheGreatestHits/class_clip_text_encodings_stable-diffusion-v1-5.pt", map_location="cpu")
        cat_map, enc_map = {"hitting with a stick": "hitting with a stick"}, {"hitting with a stick": enc}
    elif dataset in ("Landscapes", "AVSync15"):
        cat_map = json.load(open(f"./datasets/{dataset}/class_mapping.json"))
        enc_map = torch.load(f"./datasets/{dataset}/class_clip_text_encodings_stable-diffusion-v1-5.pt", map_location="cpu")
    else:
        raise Exception()
    sd15 = "./pretrained/stable-diffusion-v1-5"
    tokenizer = CLIPTokenizer.from_pretrained(sd15, subfolder="tokenizer")
    scheduler = PNDMScheduler.from_pretrained(sd15, subfolder="scheduler")
    text_encoder = CLIPTextModel.from_pretrained(sd15, subfolder="text_encoder").to(device=device, dtype=dtype)
    vae = AutoencoderKL.from_pretrained(sd15, subfolder="vae").to(device=device)
    from .audio_encoder import ImageBindSegmaskAudioEncoder

    # (:514) ImageBind-Huge audio branch from ImageBind's own checkpoint file + identity final_layer_norm, frozen
    audio_encoder = ImageBindSegmaskAudioEncoder(n_segment=video_num_frame,
                                                 imagebind_checkpoint=ImageBindSegmaskAudioEncoder.IMAGEBIND_CKPT)
    audio_encoder = audio_encoder.to(device=device).eval()
    unet = AudioUNet3DConditionModel.from_pretrained(ckpt, subfolder="unet").to(device=device)
    pipe = AudioCondAnimationPipeline(text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, scheduler=scheduler, vae=vae,
                                      audio_encoder=audio_encoder, null_text_encodings_path=null_text)
    pipe.to(torch_device=device, dtype=dtype)
    pipe.set_progress_bar_config(disable=True)
    os.makedirs(save_root, exist_ok=True)                      # exist_ok: the ranks race to create it
    todo = list(zip(filenames, categories))
    for i in adist.shard_clips(len(todo), rank, world):
        filename, category = todo[i]
        generate_videos(pipe, video_path=os.path.join(video_root, filename),
                        category_text_encoding=enc_map[cat_map[category]].view(1, 77, 768), image_size=image_size,
                        video_fps=video_fps, video_num_frame=video_num_frame, num_clips_per_video=num_clips_per_video,
                        text_guidance_scale=text_guidance_scale, audio_guidance_scale=audio_guidance_scale, seed=random_seed,
                        save_template=os.path.join(save_root, os.path.splitext(filename)[0]), device=device)


# ---- synthetic driver (replaces dataset + mp4 I/O for tests and benches) -------------------------------------------------
def synthetic_clip(seed: int, video_length: int = 12, height: int = 256, width: int = 256, text_dim: int = 768,
                   audio_dim: int = 768, device="cpu") -> dict:
    """BASELINE.json cfg-1 style inputs: image latent ~ 0.18215 N(0,1), noise N(0,1), text (77, D), audio and
    null-audio (229, D) ~ N(0,1), all from torch.Generator('cpu').manual_seed(seed)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    h, w = height // 8, width // 8
    d = dict(image_latents=torch.randn(4, h, w, generator=g) * 0.18215,
             noise=torch.randn(4, video_length - 1, h, w, generator=g),
             text_encodings=torch.randn(77, text_dim, generator=g),
             audio_encodings=torch.randn(AUDIO_TOKENS, audio_dim, generator=g),
             null_audio_encodings=torch.randn(AUDIO_TOKENS, audio_dim, generator=g))
    return {k: v.to(device) for k, v in d.items()}
