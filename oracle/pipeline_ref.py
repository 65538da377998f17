"""fp32 CPU restatement of the reference's generation loop, AudioCondAnimationPipeline.__call__
(avgen/pipelines/pipeline_audio_cond_animation.py:263-375), from latents to decoded frames, over the other
oracle pieces.  The reference pipeline module itself cannot be imported here (torchvision / diffusers pipeline base
/ ImageBind missing), so this follows :234-261 (latent preparation), :330-365 (loop, audio-only guidance,
first-frame pinning) and :368-370 + :206-213 (decode post-processing) line by line in behaviour.
"""
from __future__ import annotations

import torch

from .sched_ref import RefDDIM, RefPNDM
from .unet_ref import unet_forward
from .vae_ref import vae_decode


def prepare_video_latents(image_latents, noise, init_noise_sigma=1.0):
    """:234-261 — cat[image latent (b,4,1,h,w), noise (b,4,f-1,h,w)] * sigma_0."""
    return torch.cat([image_latents.unsqueeze(2), noise], dim=2) * init_noise_sigma


def denoise(unet_sd, unet_cfg, latents, text, audio, null_audio, mask, steps, audio_guidance=4.0, scheduler="pndm",
            trace=None, text_guidance=1.0, null_text=None):
    """latents (b,4,f,h,w); text / null_text (b,77,D); audio / null_audio (b,229,D); mask (f,229) bool.  Guidance
    mixes of :349-361 with the branch orders of :150-155 / :186-194:
      audio-only  [text+null-audio, text+audio]         eps = e0 + ag (e1 - e0)
      text-only   [null-text+audio, text+audio]         eps = e0 + tg (e1 - e0)
      dual        [uncond, text+null-audio, text+audio] eps = e0 + tg (e1 - e0) + ag (e2 - e1)
    scheduler on frames 1.. (:364)."""
    sch = RefPNDM() if scheduler == "pndm" else RefDDIM()
    sch.set_timesteps(steps)
    b, _, f = latents.shape[:3]
    do_audio, do_text = audio_guidance > 1.0, text_guidance > 1.0
    k = 1 + int(do_audio) + int(do_text)
    na = null_audio.expand_as(audio) if do_audio else None
    nt = null_text.expand_as(text) if do_text else None
    if do_text and do_audio:
        txt, aud = torch.cat([nt, text, text]), torch.cat([na, na, audio])
    elif do_text:
        txt, aud = torch.cat([nt, text]), torch.cat([audio, audio])
    elif do_audio:
        txt, aud = torch.cat([text, text]), torch.cat([na, audio])
    else:
        txt, aud = text, audio
    txt, aud = txt[:, None].expand(-1, f, -1, -1), aud[:, None].expand(-1, f, -1, -1)
    m = mask[None].expand(txt.shape[0], -1, -1)
    x = latents.clone().float()
    for t in sch.timesteps:
        xin = torch.cat([x] * k)
        n = unet_forward(unet_sd, unet_cfg, xin, int(t), txt, aud, m)
        if do_text and do_audio:
            n0, n1, n2 = n.chunk(3)
            n = n0 + text_guidance * (n1 - n0) + audio_guidance * (n2 - n1)
        elif k == 2:
            n0, n1 = n.chunk(2)
            n = n0 + (text_guidance if do_text else audio_guidance) * (n1 - n0)
        x[:, :, 1:] = sch.step(n[:, :, 1:], t, x[:, :, 1:])
        if trace is not None:
            trace.append(x.clone())
    return x


def decode(vae_sd, vae_cfg, latents):
    """:368-370 + :206-213 — (b,4,f,h,w) -> (b,f,3,H,W) in [0,1]."""
    b, c, f, h, w = latents.shape
    z = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w) / vae_cfg["scaling_factor"]
    img = vae_decode(vae_sd, vae_cfg, z)
    img = (img / 2 + 0.5).clamp(0, 1)
    return img.reshape(b, f, *img.shape[1:])
