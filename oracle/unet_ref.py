"""fp32 restatement of the reference UNet forward as pure functions over a state_dict.

`sd` maps the reference's parameter names (AudioUNet3DConditionModel.state_dict()) to tensors,
`cfg` is its config dict.  Tensors are laid out as in the reference: video (B, C, F, H, W).
Follows /root/reference/avgen/models/unets/** (cited per function); the diffusers primitives
(Attention, GEGLU FeedForward, Timesteps, TimestepEmbedding) are restated from diffusers 0.29.2.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _linear(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def sinusoidal_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers get_timestep_embedding with flip_sin_to_cos=True, downscale_freq_shift=0, scale=1,
    max_period=1e4 (as configured at audio_cond_unet_3d_condition.py:243 and
    ff_spatio_audio_temp_transformer_3d.py:250): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def timestep_mlp(x, sd, p):
    """diffusers TimestepEmbedding: linear_1 -> SiLU -> linear_2."""
    return _linear(F.silu(_linear(x, sd, p + ".linear_1")), sd, p + ".linear_2")


def ff_inflated_conv3d(x, sd, p, stride=1, padding=1):
    """FFInflatedConv3d.forward, utils.py:34-57: per-frame 2-D conv, then
    y += Linear_{3C->C}(cat[y(frame 0), y(frame max(f-1,0)), y(frame f)]) at every pixel."""
    B, C, Fr, H, W = x.shape
    y = F.conv2d(x.transpose(1, 2).reshape(B * Fr, C, H, W), sd[p + ".weight"], sd[p + ".bias"], stride=stride,
                 padding=padding)
    Co, Ho, Wo = y.shape[1:]
    y = y.reshape(B, Fr, Co, Ho, Wo).permute(0, 3, 4, 1, 2)            # (B, Ho, Wo, F, Co)
    first = y[..., :1, :].expand_as(y)
    prev = torch.cat([y[..., :1, :], y[..., :-1, :]], dim=-2)
    mix = F.linear(torch.cat([first, prev, y], dim=-1), sd[p + ".conv_temp.weight"], sd[p + ".conv_temp.bias"])
    return (y + mix).permute(0, 4, 3, 1, 2)                            # (B, Co, F, Ho, Wo)


def resnet_block(x, temb, sd, p, groups, eps):
    """FFSpatioTempResnetBlock3D.forward, ff_spatio_temp_resnet_3d.py:161-191.  GroupNorm on the 5-D
    tensor pools statistics over (channels-in-group, F, H, W).  temb: (B, F, Ct)."""
    h = F.silu(F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = ff_inflated_conv3d(h, sd, p + ".conv1")
    t = _linear(F.silu(temb), sd, p + ".time_emb_proj")                # (B, F, Co)
    h = h + t.transpose(1, 2)[:, :, :, None, None]
    h = F.silu(F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = ff_inflated_conv3d(h, sd, p + ".conv2")
    if p + ".conv_shortcut.weight" in sd:
        x = ff_inflated_conv3d(x, sd, p + ".conv_shortcut", padding=0)
    return x + h                                                       # output_scale_factor == 1


def _heads_split(x, heads):
    b, n, c = x.shape
    return x.reshape(b, n, heads, c // heads).transpose(1, 2)


def attention(x, ctx, sd, p, heads, mask=None):
    """diffusers Attention + AttnProcessor2_0: to_q / to_k / to_v (no bias), SDPA with scale d^-1/2,
    to_out.0 (bias).  mask: bool broadcastable to (B, heads, Lq, Lk), True = attend."""
    q = _heads_split(_linear(x, sd, p + ".to_q"), heads)
    k = _heads_split(_linear(ctx, sd, p + ".to_k"), heads)
    v = _heads_split(_linear(ctx, sd, p + ".to_v"), heads)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    o = o.transpose(1, 2).reshape(x.shape[0], x.shape[1], -1)
    return _linear(o, sd, p + ".to_out.0")


def transformer_block(h, text, audio, audio_mask, sd, p, heads, frames):
    """BasicTransformerBlock.forward, ff_spatio_audio_temp_transformer_3d.py:278-373.
    h: (B*F, L, C); text (B*F, 77, Dt); audio (B*F, 229, Da); audio_mask (B*F, 1, 229) bool."""
    BF, L, C = h.shape
    B = BF // frames

    def ln(x, name):
        return F.layer_norm(x, (C,), sd[p + f".{name}.weight"], sd[p + f".{name}.bias"], 1e-5)

    # 1. first-frame attention (FFAttnProcessor, utils.py:111-162): K/V of every frame <- frame 0
    n = ln(h, "norm1")
    ctx = n.reshape(B, frames, L, C)[:, :1].expand(B, frames, L, C).reshape(BF, L, C)
    h = h + attention(n, ctx, sd, p + ".attn1", heads)
    # 2. audio cross-attention with the per-frame segment mask (:315-325)
    if p + ".attn_audio.to_q.weight" in sd:
        m = None if audio_mask is None else audio_mask[:, None]        # (BF, 1, 1, 229)
        h = h + attention(ln(h, "norm_audio"), audio, sd, p + ".attn_audio", heads, m)
    # 3. text cross-attention (:328-341)
    h = h + attention(ln(h, "norm2"), text, sd, p + ".attn2", heads)
    # 4. temporal attention over frames at every pixel (:346-358); residual is the un-embedded state
    pos = timestep_mlp(sinusoidal_embedding(torch.arange(frames), C), sd, p + ".pos_embedding_temp")  # (F, C)
    ht = h.reshape(B, frames, L, C).transpose(1, 2).reshape(B * L, frames, C)
    nt = ln(ht + pos[None], "norm_temp")
    ht = ht + attention(nt, nt, sd, p + ".attn_temp", heads)
    h = ht.reshape(B, L, frames, C).transpose(1, 2).reshape(BF, L, C)
    # 5. GEGLU feed-forward (:361-371; diffusers FeedForward: proj -> value * gelu_erf(gate) -> linear)
    n = ln(h, "norm3")
    val, gate = _linear(n, sd, p + ".ff.net.0.proj").chunk(2, dim=-1)
    return h + _linear(val * F.gelu(gate), sd, p + ".ff.net.2")


def transformer_3d(x, text, audio, audio_mask, sd, p, heads, groups):
    """FFSpatioAudioTempTransformer3DModel.forward, ff_spatio_audio_temp_transformer_3d.py:94-158:
    per-frame GroupNorm (eps 1e-6) -> 1x1 proj_in -> tokens -> block -> 1x1 proj_out -> + residual."""
    B, C, Fr, H, W = x.shape
    xf = x.transpose(1, 2).reshape(B * Fr, C, H, W)
    h = F.group_norm(xf, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    h = F.conv2d(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    h = h.permute(0, 2, 3, 1).reshape(B * Fr, H * W, C)
    txt = text.reshape(B * Fr, *text.shape[2:])
    aud = audio.reshape(B * Fr, *audio.shape[2:])
    am = None if audio_mask is None else audio_mask.reshape(B * Fr, 1, -1)
    h = transformer_block(h, txt, aud, am, sd, p + ".transformer_blocks.0", heads, Fr)
    h = h.reshape(B * Fr, H, W, C).permute(0, 3, 1, 2)
    h = F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (h + xf).reshape(B, Fr, C, H, W).transpose(1, 2)


def _per_block(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


def unet_forward(sd, cfg, sample, timestep, text, audio, audio_mask):
    """AudioUNet3DConditionModel.forward, audio_cond_unet_3d_condition.py:598-798.
    sample (B,4,F,H,W); timestep scalar; text (B,F,77,Dt); audio (B,F,229,Da); audio_mask (B,F,229) bool."""
    sd = {k: v.float() for k, v in sd.items()}
    sample, text, audio = sample.float(), text.float(), audio.float()
    chans = tuple(cfg["block_out_channels"])
    nblk = len(chans)
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    heads = _per_block(cfg["attention_head_dim"], nblk)      # really the head COUNT (unet_3d_blocks.py:874-877)
    layers = _per_block(cfg["layers_per_block"], nblk)
    B, _, Fr = sample.shape[:3]

    # time embedding (:657-681): sinusoid -> MLP -> repeat over frames
    t = torch.as_tensor(timestep).reshape(-1).expand(B)
    emb = timestep_mlp(sinusoidal_embedding(t, chans[0]), sd, "time_embedding")
    temb = emb[:, None, :].expand(B, Fr, emb.shape[-1])

    h = ff_inflated_conv3d(sample, sd, "conv_in")
    skips = [h]
    for i, btype in enumerate(cfg["down_block_types"]):
        p = f"down_blocks.{i}"
        for j in range(layers[i]):
            h = resnet_block(h, temb, sd, f"{p}.resnets.{j}", groups, eps)
            if "CrossAttn" in btype:
                h = transformer_3d(h, text, audio, audio_mask, sd, f"{p}.attentions.{j}", heads[i], groups)
            skips.append(h)
        if i < nblk - 1:
            h = ff_inflated_conv3d(h, sd, f"{p}.downsamplers.0.conv", stride=2)
            skips.append(h)

    h = resnet_block(h, temb, sd, "mid_block.resnets.0", groups, eps)
    h = transformer_3d(h, text, audio, audio_mask, sd, "mid_block.attentions.0", heads[-1], groups)
    h = resnet_block(h, temb, sd, "mid_block.resnets.1", groups, eps)

    rheads = heads[::-1]
    rlayers = layers[::-1]
    for i, btype in enumerate(cfg["up_block_types"]):
        p = f"up_blocks.{i}"
        for j in range(rlayers[i] + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(h, temb, sd, f"{p}.resnets.{j}", groups, eps)
            if "CrossAttn" in btype:
                h = transformer_3d(h, text, audio, audio_mask, sd, f"{p}.attentions.{j}", rheads[i], groups)
        if i < nblk - 1:
            h = F.interpolate(h, scale_factor=(1.0, 2.0, 2.0), mode="nearest")   # ff_spatio_temp_resnet_3d.py:48
            h = ff_inflated_conv3d(h, sd, f"{p}.upsamplers.0.conv")

    h = F.silu(F.group_norm(h, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
    return ff_inflated_conv3d(h, sd, "conv_out")
