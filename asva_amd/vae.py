"""AutoencoderKL (decode side) — MI355X-native host mirror of the diffusers class the reference pipeline calls
at avgen/pipelines/pipeline_audio_cond_animation.py:206-213 (`vae.decode(latents / scaling_factor).sample` over
all b*f frames at once).  Same config keys and decoder-side state_dict names as diffusers 0.29.2's SD1.5
`vae/` checkpoint; `.decode(z).sample`, `.config.scaling_factor`, `.config.block_out_channels`, `.dtype` are the
members the pipeline touches.  The encoder half (image -> latent, once per clip, the step right
before the hot path: pipeline :198-203) runs on the same kernels: `.encode(x).latent_dist`
with `.sample(generator)` / `.mode()`.

As in asva_amd.unet the modules are parameter holders; decode() drives the gfx950 kernels: implicit-GEMM 3x3
convs (nearest-2x upsample folded into the conv's gather), GroupNorm+SiLU, and the single-head mid-block
attention as two batched GEMMs around a row softmax (d = 512 is one big contraction, not a flash-tile case).
Frames are processed in chunks sized so every activation stays below the 2 GiB / 32-bit-offset limit of the
LDS-direct GEMM tiles.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch

from . import precision as P
from torch import nn

from . import ops
from .unet import FrozenConfig, _Affine, _Conv, _Linear, _Pk, _Ref
from .weights import is_twin, pack_conv1x1, pack_conv3x3, pack_linear, rest_of, subpixel_conv3x3, to_act


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class _VaeRes(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = _Affine(cin)
        self.conv1 = _Conv(cin, cout, 3)
        self.norm2 = _Affine(cout)
        self.conv2 = _Conv(cout, cout, 3)
        if cin != cout:
            self.conv_shortcut = _Conv(cin, cout, 1)


class _VaeAttn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = _Affine(c)
        self.to_q = _Linear(c, c)
        self.to_k = _Linear(c, c)
        self.to_v = _Linear(c, c)
        self.to_out = nn.ModuleList([_Linear(c, c)])


class _VaeMid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([_VaeAttn(c)])
        self.resnets = nn.ModuleList([_VaeRes(c, c), _VaeRes(c, c)])


class _VaeUpsampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Conv(c, c, 3)


class _VaeUp(nn.Module):
    def __init__(self, cin, cout, n, up):
        super().__init__()
        self.resnets = nn.ModuleList([_VaeRes(cin if j == 0 else cout, cout) for j in range(n)])
        if up:
            self.upsamplers = nn.ModuleList([_VaeUpsampler(cout)])


class _VaeDownsampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Conv(c, c, 3)


class _VaeDown(nn.Module):
    def __init__(self, cin, cout, n, down):
        super().__init__()
        self.resnets = nn.ModuleList([_VaeRes(cin if j == 0 else cout, cout) for j in range(n)])
        if down:
            self.downsamplers = nn.ModuleList([_VaeDownsampler(cout)])


class _Encoder(nn.Module):
    def __init__(self, cin, latent, ch, layers):
        super().__init__()
        ch = list(ch)
        self.conv_in = _Conv(cin, ch[0], 3)
        self.down_blocks = nn.ModuleList()
        prev = ch[0]
        for i, c in enumerate(ch):
            self.down_blocks.append(_VaeDown(prev, c, layers, i < len(ch) - 1))
            prev = c
        self.mid_block = _VaeMid(ch[-1])
        self.conv_norm_out = _Affine(ch[-1])
        self.conv_out = _Conv(ch[-1], 2 * latent, 3)


class DiagonalGaussianDistribution:
    """diffusers' posterior object: parameters (N, 2*latent, h, w) -> mean | logvar (clamped to [-30, 20])."""

    def __init__(self, mean, logvar):
        self.mean = mean
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist

    def __getitem__(self, i):
        return (self.latent_dist,)[i]


class _Decoder(nn.Module):
    def __init__(self, latent, out, ch, layers):
        super().__init__()
        rch = list(ch)[::-1]
        self.conv_in = _Conv(latent, rch[0], 3)
        self.mid_block = _VaeMid(rch[0])
        self.up_blocks = nn.ModuleList()
        prev = rch[0]
        for i, c in enumerate(rch):
            self.up_blocks.append(_VaeUp(prev, c, layers + 1, i < len(rch) - 1))
            prev = c
        self.conv_norm_out = _Affine(rch[-1])
        self.conv_out = _Conv(rch[-1], out, 3)


# older diffusers checkpoints name the mid-block attention projections differently
# the upsamplers' nearest-2x + 3x3 convolution as four per-parity 2x2 convolutions on the original image (weights.subpixel_conv3x3): 2.09 of
# the decoder's 7.47 TFLOP per 12 x 256 x 256 clip become 0.93
_SUBPIXEL_UPS = os.environ.get("AVSD_SUBPIXEL_UPS", "1") != "0"
_LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                 up_block_types=("UpDecoderBlock2D",) * 4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 act_fn="silu", latent_channels=4, norm_num_groups=32, sample_size=512, scaling_factor=0.18215,
                 force_upcast=True, **_):
        super().__init__()
        if act_fn != "silu" or any(t != "UpDecoderBlock2D" for t in up_block_types):
            raise NotImplementedError("AutoencoderKL (MI355X path): SD1.5 decoder architecture only")
        self._config = FrozenConfig(in_channels=in_channels, out_channels=out_channels, down_block_types=tuple(down_block_types),
                                    up_block_types=tuple(up_block_types), block_out_channels=tuple(block_out_channels),
                                    layers_per_block=layers_per_block, act_fn=act_fn, latent_channels=latent_channels,
                                    norm_num_groups=norm_num_groups, sample_size=sample_size, scaling_factor=scaling_factor,
                                    force_upcast=force_upcast)
        self.encoder = _Encoder(in_channels, latent_channels, block_out_channels, layers_per_block)
        self.quant_conv = _Conv(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = _Conv(latent_channels, latent_channels, 1)
        self.decoder = _Decoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        self._packed = None

    @property
    def config(self):
        return self._config

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_config(cls, config, **kw):
        args = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        args.update(kw)
        return cls(**args)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, **_):
        p = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(p, "config.json")) as f:
            model = cls.from_config(json.load(f))
        st = os.path.join(p, "diffusion_pytorch_model.safetensors")
        if os.path.isfile(st):
            from safetensors.torch import load_file

            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(p, "diffusion_pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd)
        return model.eval()

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True):
        """diffusers directory layout: config.json + diffusion_pytorch_model.safetensors (current attention names)."""
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": "AutoencoderKL", "_diffusers_version": "0.29.2"}
        cfg.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(self.config).items()})
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file

            save_file(sd, os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(sd, os.path.join(save_directory, "diffusion_pytorch_model.bin"))

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts diffusers AutoencoderKL checkpoints of either vintage: legacy attention names
        (query/key/value/proj_attn) are mapped and legacy [C, C, 1, 1] attention weights are squeezed."""
        sd = {}
        for k, v in state_dict.items():
            parts = k.split(".")
            if "attentions" in parts:
                for old, new in _LEGACY_ATTN.items():
                    if parts[-2] == old:
                        k = ".".join(parts[:-2] + [new, parts[-1]])
                if k.endswith("weight") and v.dim() == 4 and "group_norm" not in k:
                    v = v.reshape(v.shape[0], v.shape[1])
            sd[k] = v
        r = super().load_state_dict(sd, strict=strict, **kw)
        self._packed = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = None
        return r

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x: (N, 3, H, W) in [-1, 1] -> EncoderOutput(latent_dist) (pipeline :202 then does
        `.latent_dist.sample() * scaling_factor`).  H, W multiples of 8."""
        pk = self.pack()
        dev = pk.blob.device
        n, c, H, W = x.shape
        groups = self.config.norm_num_groups
        e = pk.enc
        h = ops.ncfhw_to_rows(x.to(device=dev, dtype=torch.float32).reshape(n, c, 1, H, W).contiguous(), cpad=8)
        hw = (H, W)
        h = ops.gemm(h, e.conv_in.w, bias=e.conv_in.b, mode=ops.CONV3, conv=(n, H, W, 1, 0))
        for blk in e.down:
            for r in blk.resnets:
                h = self._res(h, r, n, hw, groups)
            if blk.down is not None:      # Downsample2D(padding=0): F.pad(0,1,0,1) then stride-2 conv == top/left pad 0
                h = ops.gemm(h, blk.down.w, bias=blk.down.b, mode=ops.CONV3, conv=(n, hw[0], hw[1], 2, 0, 0))
                hw = (hw[0] // 2, hw[1] // 2)
        h = self._res(h, e.mid[0], n, hw, groups)
        h = self._mid_attention(h, e.attn, n, hw, groups)
        h = self._res(h, e.mid[1], n, hw, groups)
        a = ops.groupnorm(h, None, n, hw[0] * hw[1], groups, e.norm_out.g, e.norm_out.b, 1e-6, True)
        h = ops.gemm(a, e.conv_out.w, bias=e.conv_out.b, mode=ops.CONV3, conv=(n, hw[0], hw[1], 1, 0))
        m = ops.gemm(h, e.quant.w, bias=e.quant.b, out_f32=True)                               # quant_conv 1x1 -> moments
        lat = self.config.latent_channels
        m = ops.rows_to_ncfhw(m, n, 2 * lat, 1, hw[0], hw[1]).reshape(n, 2 * lat, hw[0], hw[1])
        dist = DiagonalGaussianDistribution(m[:, :lat].contiguous(), m[:, lat:].contiguous())
        return EncoderOutput(dist) if return_dict else (dist,)

    # ---- packing -----------------------------------------------------------------------------------------------
    def pack(self, device=None):
        if device is not None:
            device = torch.device(device)
            if device.type == "cuda" and device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
        if (self._packed is not None and (device is None or self._packed.blob.device == device)
                and getattr(self._packed, "split", False) == P.SPLIT and getattr(self._packed, "act_dtype", P.ACT) == P.ACT):
            return self._packed
        device = device if device is not None else self.device
        if device.type != "cuda" and not getattr(ops, "EMULATED", False):
            raise RuntimeError("AutoencoderKL.pack: the MI355X path needs a cuda (HIP) device — there is no CPU compute path")
        items = []

        def reg(t):
            items.append(t.contiguous())
            return _Ref(len(items) - 1)

        def aff(m):
            return _Pk(g=reg(m.weight.detach().float()), b=reg(m.bias.detach().float()))

        def conv3(m):
            cout, cin = m.weight.shape[:2]
            cop, cip = (cout + 3) // 4 * 4, (cin + 7) // 8 * 8
            b = torch.zeros(cop, device=m.weight.device)
            b[:cout] = m.bias.detach().float()
            return _Pk(w=reg(pack_conv3x3(m.weight.detach().float(), cip, cop)), b=reg(b), cout=cop)

        def conv3_up(m):
            """an upsampler's convolution (diffusers Upsample2D: nearest 2x, then 3x3 pad 1) in its sub-pixel form — four per-parity 2x2
            kernels on the original image, 4/9 of the multiplies (weights.subpixel_conv3x3; AVSD_GEMM_CONV3 with ups = 2) — where the
            kernel's layout rules hold (whole 64-channel K tiles per tap, whole column tiles per parity)"""
            cout, cin = m.weight.shape[:2]
            if not _SUBPIXEL_UPS or cout % 64 or cin % 64:
                return conv3(m)
            wf = m.weight.detach().float().permute(0, 2, 3, 1).contiguous()
            return _Pk(w=reg(to_act(subpixel_conv3x3(wf))), b=reg(m.bias.detach().float().repeat(4)), cout=cout, subpixel=True)

        def conv1(m):
            cout, cin = m.weight.shape[:2]
            cop, cip = (cout + 7) // 8 * 8, (cin + 7) // 8 * 8
            w = torch.zeros(cop, cip, device=m.weight.device)
            w[:cout, :cin] = m.weight.detach().float().reshape(cout, cin)
            b = torch.zeros(cop, device=m.weight.device)
            b[:cout] = m.bias.detach().float()
            return _Pk(w=reg(to_act(w)), b=reg(b))

        def res(m):
            return _Pk(norm1=aff(m.norm1), conv1=conv3(m.conv1), norm2=aff(m.norm2), conv2=conv3(m.conv2),
                       shortcut=conv1(m.conv_shortcut) if hasattr(m, "conv_shortcut") else None)

        def attn(at):
            return _Pk(norm=aff(at.group_norm),
                       wqk=reg(pack_linear(torch.cat([at.to_q.weight, at.to_k.weight], 0).float())),
                       bqk=reg(torch.cat([at.to_q.bias, at.to_k.bias], 0).detach().float()),
                       wv=reg(pack_linear(at.to_v.weight.float())), bv=reg(at.to_v.bias.detach().float()),
                       wqkv=reg(pack_linear(torch.cat([at.to_q.weight, at.to_k.weight, at.to_v.weight], 0).float())),
                       bqkv=reg(torch.cat([at.to_q.bias, at.to_k.bias, at.to_v.bias], 0).detach().float()),
                       wo=reg(pack_linear(at.to_out[0].weight.float())), bo=reg(at.to_out[0].bias.detach().float()),
                       dim=at.to_q.weight.shape[0])

        d = self.decoder
        en = self.encoder
        enc = _Pk(conv_in=conv3(en.conv_in),
                  down=[_Pk(resnets=[res(r) for r in b.resnets],
                            down=conv3(b.downsamplers[0].conv) if hasattr(b, "downsamplers") else None) for b in en.down_blocks],
                  mid=[res(en.mid_block.resnets[0]), res(en.mid_block.resnets[1])], attn=attn(en.mid_block.attentions[0]),
                  norm_out=aff(en.conv_norm_out), conv_out=conv3(en.conv_out), quant=conv1(self.quant_conv))
        pk = _Pk(enc=enc, pq=conv1(self.post_quant_conv), conv_in=conv3(d.conv_in),
                 mid=[res(d.mid_block.resnets[0]), res(d.mid_block.resnets[1])],
                 attn=attn(d.mid_block.attentions[0]),
                 up=[_Pk(resnets=[res(r) for r in u.resnets],
                         up=conv3_up(u.upsamplers[0].conv) if hasattr(u, "upsamplers") else None) for u in d.up_blocks],
                 norm_out=aff(d.conv_norm_out), conv_out=conv3(d.conv_out))
        offs, total = [], 0
        for t in items:
            offs.append(total)
            total += (t.numel() * t.element_size() + 255) // 256 * 256
        # split precision: a twin blob, the rest plane of every 16-bit item at the same offset in the second half (precision.py)
        blob = torch.zeros(total * (2 if P.SPLIT else 1), dtype=torch.uint8, device=device)
        views = []
        for t, o in zip(items, offs):
            nb = t.numel() * t.element_size()
            if not t.is_meta:
                blob[o:o + nb].copy_(t.reshape(-1).view(torch.uint8))
                if P.SPLIT and t.dtype == P.ACT:
                    if not is_twin(t):
                        raise RuntimeError("split-precision packing: a 16-bit item was not produced by weights.to_act")
                    blob[total + o:total + o + nb].copy_(rest_of(t).reshape(-1).view(torch.uint8))
            views.append(blob[o:o + nb].view(t.dtype).view(t.shape))

        def resolve(obj):
            if isinstance(obj, _Pk):
                for k, v in list(obj.__dict__.items()):
                    if isinstance(v, _Ref):
                        obj.__dict__[k] = views[v.idx]
                    else:
                        resolve(v)
            elif isinstance(obj, list):
                for v in obj:
                    resolve(v)

        resolve(pk)
        pk.blob = blob
        pk.split = P.SPLIT
        pk.act_dtype = P.ACT
        self._packed = pk
        return pk

    # ---- decode ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, frames_per_chunk: Optional[int] = None,
               postprocess=False):
        """z: (N, 4, h, w) latents already divided by scaling_factor -> DecoderOutput(sample (N, 3, 8h, 8w) f32).
        postprocess=True additionally applies the pipeline's (x / 2 + 0.5).clamp(0, 1) in the output kernel;
        postprocess="uint8" returns uint8 frames (N, H, W, 3) = trunc(that * 255), ready for the video writer."""
        pk = self.pack()
        dev = pk.blob.device
        N, _, h, w = z.shape
        up = 2 ** (len(self.config.block_out_channels) - 1)
        if frames_per_chunk is None:
            # largest activation: rows x 256 ch 16-bit at full resolution (the mid attention is flash-style at C = 512; the
            # explicit form of other channel counts materialises f32 HW x HW scores)
            mid_c = self.config.block_out_channels[-1]
            per_frame = max((h * up) * (w * up) * 256 * 2, 0 if mid_c == 512 else (h * w) ** 2 * 4)
            frames_per_chunk = max(1, min(N, int((2 ** 31 - 1) // per_frame)))
        outs = []
        z32 = z.to(device=dev, dtype=torch.float32).contiguous()
        for i in range(0, N, frames_per_chunk):
            outs.append(self._decode_rows(pk, z32[i:i + frames_per_chunk], postprocess))
        img = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
        if not return_dict:
            return (img,)
        return DecoderOutput(img)

    def decode_to_video(self, latents: torch.Tensor) -> torch.Tensor:
        """(b, 4, f, h, w) denoised latents -> (b, f, 3, H, W) f32 in [0, 1]: the whole post-processing of
        pipeline_audio_cond_animation.py:368-370 + :207-212 (1/scaling_factor, decode, x/2+0.5, clamp)."""
        b, c, f, h, w = latents.shape
        img = self._decode_clip(latents, True)
        if img is None:
            z = latents.to(torch.float32).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w) / self.config.scaling_factor
            img = self.decode(z, postprocess=True).sample
        return img.reshape(b, f, *img.shape[1:])

    def decode_to_uint8_frames(self, latents: torch.Tensor) -> torch.Tensor:
        """(b, 4, f, h, w) denoised latents -> (b, f, H, W, 3) uint8 on the device (what generate_videos hands to
        the video writer, pipeline :448), without the f32 round trip."""
        b, c, f, h, w = latents.shape
        img = self._decode_clip(latents, "uint8")
        if img is None:
            z = latents.to(torch.float32).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w) / self.config.scaling_factor
            img = self.decode(z, postprocess="uint8").sample
        return img.reshape(b, f, *img.shape[1:])

    def _decode_clip(self, latents: torch.Tensor, postprocess):
        """All frames of the clips in one pass, straight from the (b, 4, f, h, w) latents — every device operation is a
        library launch (recordable as a launch plan, asva_amd/plan.py).  None when the frames need chunking."""
        pk = self.pack()
        b, c, f, h, w = latents.shape
        up = 2 ** (len(self.config.block_out_channels) - 1)
        mid_c = self.config.block_out_channels[-1]
        per_frame = max((h * up) * (w * up) * 256 * 2, 0 if mid_c == 512 else (h * w) ** 2 * 4)
        if b * f * per_frame > 2 ** 31 - 1:
            return None
        lat32 = latents.to(device=pk.blob.device, dtype=torch.float32).contiguous()
        return self._decode_rows(pk, None, postprocess, latents5=lat32)

    def _res(self, x, p, n, hw, groups):
        L = hw[0] * hw[1]
        a = ops.groupnorm(x, None, n, L, groups, p.norm1.g, p.norm1.b, 1e-6, True)
        h = ops.gemm(a, p.conv1.w, bias=p.conv1.b, mode=ops.CONV3, conv=(n, hw[0], hw[1], 1, 0))
        a = ops.groupnorm(h, None, n, L, groups, p.norm2.g, p.norm2.b, 1e-6, True)
        s = x if p.shortcut is None else ops.gemm(x, p.shortcut.w, bias=p.shortcut.b)
        return ops.gemm(a, p.conv2.w, bias=p.conv2.b, res1=s, mode=ops.CONV3, conv=(n, hw[0], hw[1], 1, 0))

    def _mid_attention(self, x, a, n, hw, groups):
        """mid-block attention, one head of width C (diffusers UNetMidBlock2D: softmax(Q K^T / sqrt(C)) V per image): one
        fused q|k|v projection, then the single-wide-head flash kernel (attn_wide_kernel: C = 512 split over the waves of a
        workgroup) — no L x L score matrix in memory.  Channel counts that kernel is not built for (tiny test
        configurations) take the explicit form: S = QK^T/sqrt(C) (f32) -> softmax -> P V; V^T comes straight out of a GEMM
        with the roles swapped, and V's bias is added after P.V (softmax rows sum to 1)."""
        C, L = a.dim, hw[0] * hw[1]
        xn = ops.groupnorm(x, None, n, L, groups, a.norm.g, a.norm.b, 1e-6, False)
        if C == 512:
            qkv = ops.gemm(xn, a.wqkv, bias=a.bqkv)                                                   # [n*L, 3C]
            o = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], bq=n, lq=L, lk=L, kv_rows=L, heads=1, q_per_kv=1, frames=1)
            return ops.gemm(o, a.wo, bias=a.bo, res1=x)
        qk = ops.gemm(xn, a.wqk, bias=a.bqk).view(n, L, 2 * C)
        s = ops.gemm_batched(qk[:, :, :C], qk[:, :, C:], alpha=float(C) ** -0.5, out_f32=True)      # [n, L, L]
        p = ops.softmax_rows(s.view(n * L, L)).view(n, L, L)
        vt = ops.gemm_batched(a.wv.unsqueeze(0).expand(n, C, C), xn.view(n, L, C))                  # [n, C, L] = V^T
        o = ops.gemm_batched(p, vt, bias=a.bv).view(n * L, C)
        return ops.gemm(o, a.wo, bias=a.bo, res1=x)

    def _decode_rows(self, pk, z32, postprocess=False, latents5=None):
        """z32 (n, 4, h, w) f32, or latents5 (b, 4, f, h, w) f32 still multiplied by scaling_factor: the layout kernel then also
        does the pipeline's permute + 1/scaling_factor (:207-209), and image b*f + i is frame i of clip b"""
        groups = self.config.norm_num_groups
        if latents5 is not None:
            b, c, f, h, w = latents5.shape
            n = b * f
            x = ops.ncfhw_to_rows(latents5, cpad=8, scale=1.0 / self.config.scaling_factor)
        else:
            n, c, h, w = z32.shape
            x = ops.ncfhw_to_rows(z32.reshape(n, c, 1, h, w), cpad=8)
        x = ops.gemm(x, pk.pq.w, bias=pk.pq.b)                                            # post_quant_conv 1x1
        hw = (h, w)
        x = ops.gemm(x, pk.conv_in.w, bias=pk.conv_in.b, mode=ops.CONV3, conv=(n, h, w, 1, 0))
        x = self._res(x, pk.mid[0], n, hw, groups)
        x = self._mid_attention(x, pk.attn, n, hw, groups)
        x = self._res(x, pk.mid[1], n, hw, groups)
        for u in pk.up:
            for r in u.resnets:
                x = self._res(x, r, n, hw, groups)
            if u.up is not None:
                x = ops.gemm(x, u.up.w, bias=u.up.b, mode=ops.CONV3, conv=(n, hw[0], hw[1], 1, 2 if getattr(u.up, "subpixel", False) else 1))
                hw = (hw[0] * 2, hw[1] * 2)
        a = ops.groupnorm(x, None, n, hw[0] * hw[1], groups, pk.norm_out.g, pk.norm_out.b, 1e-6, True)
        if postprocess:
            y = ops.gemm(a, pk.conv_out.w, bias=pk.conv_out.b, mode=ops.CONV3, conv=(n, hw[0], hw[1], 1, 0))
            if postprocess == "uint8":
                return ops.vae_postprocess_u8(y, n, hw[0], hw[1])
            return ops.vae_postprocess(y, n, hw[0], hw[1])
        y = ops.gemm(a, pk.conv_out.w, bias=pk.conv_out.b, out_f32=True, mode=ops.CONV3, conv=(n, hw[0], hw[1], 1, 0))
        return ops.rows_to_ncfhw(y, n, self.config.out_channels, 1, hw[0], hw[1]).reshape(n, self.config.out_channels, hw[0], hw[1])
