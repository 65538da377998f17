"""Generates tests/golden/* by running the REFERENCE's own UNet code (imported from /root/reference)
on CPU in the build container.  Not runnable on the GPU box (no /root/reference there) and never
imported by tests: the committed outputs are the fixtures.

    python oracle/gen_golden.py [--full]      # --full also runs the 1.17 B-parameter SD1.5-shaped model

The reference needs diffusers==0.29.2, absent here: oracle/diffusers_restated supplies the few symbols
it imports (see that package's docstring).  Weights come from oracle/filler.py (closed form), so only
inputs and outputs are stored.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "diffusers_restated"), "/root/reference", ROOT]

from avgen.models.unets import AudioUNet3DConditionModel  # noqa: E402  (the reference)
from avgen.models.unets.resnets.ff_spatio_temp_resnet_3d import (  # noqa: E402
    FFSpatioTempResDownsample3D, FFSpatioTempResnetBlock3D, FFSpatioTempResUpsample3D)
from avgen.models.unets.transformers.ff_spatio_audio_temp_transformer_3d import (  # noqa: E402
    BasicTransformerBlock, FFSpatioAudioTempTransformer3DModel)
from avgen.models.unets.utils import FFAttention, FFInflatedConv3d  # noqa: E402

from asva_amd.conditioning import audio_segment_mask  # noqa: E402
from oracle.filler import fill_module_, seeded_randn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

TINY_CFG = dict(block_out_channels=(80, 160, 160, 160), attention_head_dim=2, norm_num_groups=16,
                cross_attention_dim=64, audio_cross_attention_dim=64, sample_size=8)
SD15_CFG = dict(block_out_channels=(320, 640, 1280, 1280), attention_head_dim=8, norm_num_groups=32,
                cross_attention_dim=768, audio_cross_attention_dim=768, sample_size=32)


def jsonable(cfg):
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(cfg).items()}


@torch.no_grad()
def tiny_e2e():
    m = AudioUNet3DConditionModel(**TINY_CFG).eval()
    fill_module_(m)
    B, Fr, H, W = 2, 4, 8, 8
    x = seeded_randn(1, B, 4, Fr, H, W)
    text = seeded_randn(2, B, 1, 7, 64).expand(B, Fr, 7, 64).contiguous()
    audio = seeded_randn(3, B, 1, 229, 64).expand(B, Fr, 229, 64).contiguous()
    mask = audio_segment_mask(Fr)[None].expand(B, -1, -1).contiguous()
    outs = {}
    for t in (981, 1):
        outs[t] = m(x, torch.tensor(t), text, audio, audio_attention_mask=mask).sample
    torch.save({"config": jsonable(m.config), "sample": x, "text": text[:, 0].clone(), "audio": audio[:, 0].clone(),
                "mask": mask[0].clone(), "timesteps": [981, 1], "out": [outs[981], outs[1]]},
               os.path.join(OUT, "unet_tiny_e2e.pt"))
    json.dump({k: list(v.shape) for k, v in m.state_dict().items()},
              open(os.path.join(OUT, "unet_tiny_state_dict_shapes.json"), "w"), indent=0)
    print("tiny e2e: out std", outs[981].std().item(), "params", sum(p.numel() for p in m.parameters()))


@torch.no_grad()
def per_op():
    g = {}
    B, Fr = 2, 3
    # FFInflatedConv3d: 3x3 stride 1, 3x3 stride 2, 1x1 (utils.py:22-57)
    for name, (cin, cout, k, s, p) in {"conv3_s1": (16, 24, 3, 1, 1), "conv3_s2": (16, 16, 3, 2, 1), "conv1": (24, 16, 1, 1, 0)}.items():
        m = FFInflatedConv3d(cin, cout, k, stride=s, padding=p)
        fill_module_(m, name + ".")
        x = seeded_randn(10, B, cin, Fr, 6, 8)
        g[name] = {"x": x, "y": m(x), "args": (cin, cout, k, s, p)}
    # FFAttention / FFAttnProcessor (utils.py:60-162): K/V from frame 0
    m = FFAttention(query_dim=32, heads=4, dim_head=8)
    fill_module_(m, "ffattn.")
    x = seeded_randn(11, B * Fr, 10, 32)
    g["ffattn"] = {"x": x, "y": m(x, video_length=Fr), "heads": 4, "frames": Fr}
    # ResBlock (ff_spatio_temp_resnet_3d.py:99-191), with and without shortcut
    for name, (cin, cout) in {"res_same": (32, 32), "res_short": (48, 32)}.items():
        m = FFSpatioTempResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=64, groups=8, eps=1e-5)
        fill_module_(m, name + ".")
        x = seeded_randn(12, B, cin, Fr, 6, 8)
        temb = seeded_randn(13, B, 1, 64).expand(B, Fr, 64).contiguous()
        g[name] = {"x": x, "temb": temb, "y": m(x, temb), "groups": 8, "eps": 1e-5}
    # down / up samplers (:10-96)
    m = FFSpatioTempResDownsample3D(16, use_conv=True, out_channels=16, padding=1, name="op")
    fill_module_(m, "down.")
    x = seeded_randn(14, B, 16, Fr, 8, 8)
    g["down"] = {"x": x, "y": m(x)}
    m = FFSpatioTempResUpsample3D(16, use_conv=True, out_channels=16)
    fill_module_(m, "up.")
    g["up"] = {"x": x, "y": m(x)}
    # full transformer wrapper + block (ff_spatio_audio_temp_transformer_3d.py:33-373)
    m = FFSpatioAudioTempTransformer3DModel(4, 8, in_channels=32, num_layers=1, cross_attention_dim=24,
                                            audio_cross_attention_dim=40, norm_num_groups=8)
    fill_module_(m, "tr.")
    x = seeded_randn(15, B, 32, Fr, 4, 6)
    text = seeded_randn(16, B, 1, 7, 24).expand(B, Fr, 7, 24).contiguous()
    audio = seeded_randn(17, B, 1, 229, 40).expand(B, Fr, 229, 40).contiguous()
    mask = audio_segment_mask(Fr)[None].expand(B, -1, -1).contiguous()
    y = m(x, encoder_hidden_states=text, audio_encoder_hidden_states=audio, audio_attention_mask=mask).sample
    g["transformer3d"] = {"x": x, "text": text, "audio": audio, "mask": mask, "y": y, "heads": 4, "groups": 8}
    torch.save(g, os.path.join(OUT, "unet_ops.pt"))
    print("per-op goldens:", list(g))


@torch.no_grad()
def full_shape():
    torch.set_num_threads(os.cpu_count())
    m = AudioUNet3DConditionModel(**SD15_CFG).eval()
    shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
    json.dump(shapes, open(os.path.join(OUT, "unet_sd15_state_dict_shapes.json"), "w"), indent=0)
    json.dump(jsonable(m.config), open(os.path.join(OUT, "unet_sd15_config.json"), "w"), indent=1)
    fill_module_(m)
    B, Fr, H, W = 2, 12, 32, 32
    lat = seeded_randn(1, 1, 4, Fr, H, W)
    x = torch.cat([lat, lat])                                   # CFG duplication
    text = seeded_randn(2, 1, 77, 768).expand(2, 77, 768)
    audio = torch.cat([seeded_randn(4, 1, 229, 768), seeded_randn(3, 1, 229, 768)])   # [null-audio, audio]
    mask = audio_segment_mask(Fr)[None].expand(B, -1, -1).contiguous()
    y = m(x, torch.tensor(981), text[:, None].expand(B, Fr, 77, 768), audio[:, None].expand(B, Fr, 229, 768),
          audio_attention_mask=mask).sample
    flat = y.flatten()
    idx = torch.arange(0, flat.numel(), flat.numel() // 4096)[:4096]
    torch.save({"timestep": 981, "idx": idx, "sample": flat[idx].clone(), "mean": flat.mean().item(),
                "std": flat.std().item(), "absmax": flat.abs().max().item(), "norm": flat.norm().item(),
                "shape": list(y.shape), "full": y.to(torch.float16)},
               os.path.join(OUT, "unet_sd15_forward.pt"))
    print("full-shape forward: std", flat.std().item(), "params", sum(p.numel() for p in m.parameters()), len(shapes))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    tiny_e2e()
    per_op()
    if a.full:
        full_shape()
