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
            trace=None):
    """latents (b,4,f,h,w); text (b,77,D); audio / null_audio (b,229,D); mask (f,229) bool.  Audio-only guidance:
    UNet batch [text+null-audio, text+audio] (:155,:193-194), eps = e0 + g (e1 - e0) (:358-361), scheduler on frames
    1.. (:364)."""
    sch = RefPNDM() if scheduler == "pndm" else RefDDIM()
    sch.set_timesteps(steps)
    b, _, f = latents.shape[:3]
    do_cfg = audio_guidance > 1.0
    txt = (torch.cat([text, text]) if do_cfg else text)[:, None].expand(-1, f, -1, -1)
    aud = (torch.cat([null_audio.expand_as(audio), audio]) if do_cfg else audio)[:, None].expand(-1, f, -1, -1)
    m = mask[None].expand(txt.shape[0], -1, -1)
    x = latents.clone().float()
    for t in sch.timesteps:
        xin = torch.cat([x, x]) if do_cfg else x
        n = unet_forward(unet_sd, unet_cfg, xin, int(t), txt, aud, m)
        if do_cfg:
            n0, n1 = n.chunk(2)
            n = n0 + audio_guidance * (n1 - n0)
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
