"""fp32 restatement of AutoencoderKL.decode (diffusers 0.29.2, SD1.5 `vae/config.json`) as called by the
reference at avgen/pipelines/pipeline_audio_cond_animation.py:206-213 (`vae.decode(latents / 0.18215).sample`).

The VAE source is NOT in /root/reference (third-party, not vendored, not installed): this follows the published
architecture — post_quant_conv 1x1 -> Decoder: conv_in, mid block (ResnetBlock2D, single-head attention with
GroupNorm, ResnetBlock2D), 4 UpDecoderBlock2D of 3 resnets [512, 512, 256, 128] with nearest-2x + 3x3 conv after
the first three, GroupNorm(32, eps 1e-6) -> SiLU -> conv_out.  PARITY UNPINNED by reference-side vectors; every
primitive is a torch built-in (F.conv2d, F.group_norm, F.scaled_dot_product_attention, F.interpolate).
State-dict names are diffusers' (`decoder.up_blocks.2.resnets.0.conv_shortcut.weight`, ...).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SD15_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, act_fn="silu", sample_size=512)


def _conv(x, sd, p, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def resnet(x, sd, p, groups, eps=1e-6):
    h = F.silu(F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = _conv(h, sd, p + ".conv1", 1)
    h = F.silu(F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = _conv(h, sd, p + ".conv2", 1)
    if p + ".conv_shortcut.weight" in sd:
        x = _conv(x, sd, p + ".conv_shortcut")
    return x + h


def mid_attention(x, sd, p, groups, eps=1e-6):
    """Attention(heads=1, dim_head=C, bias=True, residual_connection=True, norm_num_groups=32)."""
    B, C, H, W = x.shape
    h = F.group_norm(x.reshape(B, C, H * W), groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], eps)
    h = h.transpose(1, 2)                                                  # (B, HW, C)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])[:, None]
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])[:, None]
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])[:, None]
    o = F.scaled_dot_product_attention(q, k, v)[:, 0]
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def vae_decode(sd, cfg, z):
    """z: (N, 4, h, w) ALREADY divided by scaling_factor (the caller does that, pipeline :210). -> (N, 3, 8h, 8w)."""
    sd = {k: v.float() for k, v in sd.items()}
    groups = cfg["norm_num_groups"]
    nblk = len(cfg["block_out_channels"])
    h = _conv(z.float(), sd, "post_quant_conv")
    h = _conv(h, sd, "decoder.conv_in", 1)
    h = resnet(h, sd, "decoder.mid_block.resnets.0", groups)
    h = mid_attention(h, sd, "decoder.mid_block.attentions.0", groups)
    h = resnet(h, sd, "decoder.mid_block.resnets.1", groups)
    for i in range(nblk):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet(h, sd, f"decoder.up_blocks.{i}.resnets.{j}", groups)
        if i < nblk - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", 1)
    h = F.silu(F.group_norm(h, groups, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    return _conv(h, sd, "decoder.conv_out", 1)


def vae_encode_moments(sd, cfg, x):
    """AutoencoderKL.encode up to the posterior parameters (pipeline :198-203 calls `.latent_dist.sample()` on it):
    Encoder = conv_in -> 4 DownEncoderBlock2D (2 resnets each; Downsample2D = F.pad(0,1,0,1) + 3x3 stride-2 conv,
    padding 0, after the first three) -> mid block (resnet, attention, resnet) -> GroupNorm(eps 1e-6) -> SiLU ->
    conv_out (2*latent channels) -> quant_conv 1x1.  x: (N, 3, H, W) in [-1, 1] -> (mean, logvar) each (N, 4, H/8, W/8);
    logvar clamped to [-30, 20] (DiagonalGaussianDistribution)."""
    sd = {k: v.float() for k, v in sd.items()}
    groups = cfg["norm_num_groups"]
    nblk = len(cfg["block_out_channels"])
    h = _conv(x.float(), sd, "encoder.conv_in", 1)
    for i in range(nblk):
        for j in range(cfg["layers_per_block"]):
            h = resnet(h, sd, f"encoder.down_blocks.{i}.resnets.{j}", groups)
        if i < nblk - 1:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2)
    h = resnet(h, sd, "encoder.mid_block.resnets.0", groups)
    h = mid_attention(h, sd, "encoder.mid_block.attentions.0", groups)
    h = resnet(h, sd, "encoder.mid_block.resnets.1", groups)
    h = F.silu(F.group_norm(h, groups, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6))
    h = _conv(h, sd, "encoder.conv_out", 1)
    m = _conv(h, sd, "quant_conv")
    mean, logvar = m.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def encoder_shapes(cfg) -> dict:
    """name -> shape of the encoder-side state_dict (encoder.* + quant_conv)."""
    ch = list(cfg["block_out_channels"])
    lat = cfg["latent_channels"]
    s = {}

    def conv(p, ci, co, k):
        s[p + ".weight"] = (co, ci, k, k)
        s[p + ".bias"] = (co,)

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", ci, co, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", ci, co, 1)

    conv("encoder.conv_in", cfg["in_channels"], ch[0], 3)
    prev = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg["layers_per_block"]):
            res(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        if i < len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
        prev = co
    top = ch[-1]
    res("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{a}.{n}.weight"] = (top, top)
        s[f"{a}.{n}.bias"] = (top,)
    res("encoder.mid_block.resnets.1", top, top)
    norm("encoder.conv_norm_out", top)
    conv("encoder.conv_out", top, 2 * lat, 3)
    conv("quant_conv", 2 * lat, 2 * lat, 1)
    return s


def decoder_shapes(cfg) -> dict:
    """name -> shape of the decoder-side state_dict (post_quant_conv + decoder.*), from the architecture."""
    ch = list(cfg["block_out_channels"])
    lat, out = cfg["latent_channels"], cfg["out_channels"]
    s = {}

    def conv(p, ci, co, k):
        s[p + ".weight"] = (co, ci, k, k)
        s[p + ".bias"] = (co,)

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", ci, co, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", ci, co, 1)

    conv("post_quant_conv", lat, lat, 1)
    top = ch[-1]
    conv("decoder.conv_in", lat, top, 3)
    res("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[f"{a}.{n}.weight"] = (top, top)
        s[f"{a}.{n}.bias"] = (top,)
    res("decoder.mid_block.resnets.1", top, top)
    rch = ch[::-1]
    prev = rch[0]
    for i, co in enumerate(rch):
        for j in range(cfg["layers_per_block"] + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        if i < len(rch) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        prev = co
    norm("decoder.conv_norm_out", rch[-1])
    conv("decoder.conv_out", rch[-1], out, 3)
    return s
