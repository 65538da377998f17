"""ImageBindSegmaskAudioEncoder on the HIP kernels — host mirror of
avgen/models/audio_encoders/segmask_imagebind.py:44-123.

The reference wraps the audio branch of ImageBind-Huge (`imagebind_model.imagebind_huge(pretrained=True)`: modality
preprocessor, trunk, head, postprocessor) and adds a trainable `final_layer_norm` plus the per-frame segment masks.
ImageBind is an un-vendored submodule (facebookresearch/ImageBind, README.md:61; no pinned commit in the reference
tree), so the audio branch is restated from its published architecture — parity unpinned:

  preprocessor  AudioPreprocessor: PatchEmbedGeneric(Conv2d(1, 768, kernel 16, stride 10, bias=False) + LayerNorm(768))
                on the (1, 128, 204) mel-spectrogram -> 12 x 19 = 228 patches, one cls token, learned position table
                (1, 229, 768)
  trunk         SimpleTransformer: 12 pre-LN blocks (LayerNorm eps 1e-6), torch.nn.MultiheadAttention(768, 12 heads,
                bias=True, add_bias_kv=True) — one learned key/value pair appended after the 229 tokens — and a
                GELU MLP 768 -> 3072 -> 768; drop-path is identity at inference; no final norm in the trunk
  head          LayerNorm(eps 1e-6) -> cls token -> Linear(768, 1024, bias=False); postprocessor = L2 normalise +
                fixed logit scale 20 (only with normalize=True, which the pipeline never asks for)

Parameter names follow the ImageBind modules so the `modules/audio_encoder` checkpoint the reference trainer writes
(audio_cond_animation_trainer.py:152-155) loads as is.  All arithmetic runs in libavsd_hip.so: patch gather,
GEMMs (bias / GELU / residual epilogues), LayerNorm, attention (d = 64, 230 keys).  The token sequences carry one
extra row per clip: after the in-projection its key/value slots are overwritten with bias_k / bias_v, which is exactly
the appended pair; its query/output are computed and never read.
"""
from __future__ import annotations

import json
import math
import os
from typing import Any, Dict, Optional

import torch

from . import precision as P
import torch.nn as nn

from . import ops
from .conditioning import audio_segment_mask
from .unet import BIN_NAME, CONFIG_NAME, SAFETENSORS_NAME, FrozenConfig

EMBED, HEADS, DEPTH, MLP, OUT_DIM = 768, 12, 12, 3072, 1024
MEL, FRAMES, PATCH, STRIDE = 128, 204, 16, 10
N_FREQ, N_TIME = (MEL - PATCH) // STRIDE + 1, (FRAMES - PATCH) // STRIDE + 1      # 12 x 19 (segmask_imagebind.py:104)


class ImageBindSegmaskAudioEncoderOutput(dict):
    """audio_embeds / audio_encodings / audio_segment_masks (segmask_imagebind.py:21-41)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to_tuple(self):
        return tuple(self.values())


# ---- parameter holders (never called; shapes and names of the ImageBind modules) ---------------------------------
class _P(nn.Module):
    def __init__(self, **shapes):
        super().__init__()
        for k, (shape, std) in shapes.items():
            t = torch.zeros(shape)
            if std == "ones":
                t.fill_(1.0)
            elif std:
                t.normal_(0.0, std)
            self.register_parameter(k, nn.Parameter(t, requires_grad=False))


class _Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.proj = _P(weight=((EMBED, 1, PATCH, PATCH), 0.02))
        self.norm_layer = _P(weight=((EMBED,), "ones"), bias=((EMBED,), 0))


class _Preprocessor(nn.Module):
    def __init__(self):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, EMBED).normal_(0.0, EMBED ** -0.5), requires_grad=False)
        self.rgbt_stem = _Stem()
        self.pos_embedding_helper = _P(pos_embed=((1, 1 + N_FREQ * N_TIME, EMBED), 0.02))


class _Attn(nn.Module):
    def __init__(self):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.zeros(3 * EMBED, EMBED).normal_(0.0, 0.02), requires_grad=False)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * EMBED), requires_grad=False)
        self.bias_k = nn.Parameter(torch.zeros(1, 1, EMBED).normal_(0.0, 0.02), requires_grad=False)
        self.bias_v = nn.Parameter(torch.zeros(1, 1, EMBED).normal_(0.0, 0.02), requires_grad=False)
        self.out_proj = _P(weight=((EMBED, EMBED), 0.02), bias=((EMBED,), 0))


class _Mlp(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = _P(weight=((MLP, EMBED), 0.02), bias=((MLP,), 0))
        self.fc2 = _P(weight=((EMBED, MLP), 0.02), bias=((EMBED,), 0))


class _BlockP(nn.Module):
    def __init__(self):
        super().__init__()
        self.attn = _Attn()
        self.norm_1 = _P(weight=((EMBED,), "ones"), bias=((EMBED,), 0))
        self.mlp = _Mlp()
        self.norm_2 = _P(weight=((EMBED,), "ones"), bias=((EMBED,), 0))


class _Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([_BlockP() for _ in range(DEPTH)])


class ImageBindSegmaskAudioEncoder(nn.Module):
    config_name = CONFIG_NAME

    IMAGEBIND_CKPT = ".checkpoints/imagebind_huge.pth"      # where imagebind_huge(pretrained=True) keeps its weights

    def __init__(self, n_segment: int = 4, pretrained_model_name: str = "imagebind-huge", imagebind_checkpoint: Optional[str] = None):
        """imagebind_checkpoint: path of ImageBind's own `imagebind_huge.pth` (the reference constructor loads it through
        `imagebind_huge(pretrained=True)`, segmask_imagebind.py:56); None keeps the random initialisation (synthetic
        benchmarks / tests — there is no network to fetch the file)."""
        super().__init__()
        if pretrained_model_name != "imagebind-huge":
            raise NotImplementedError(pretrained_model_name)
        object.__setattr__(self, "config", FrozenConfig(n_segment=n_segment, pretrained_model_name=pretrained_model_name))
        self.n_segment = n_segment
        self.pretrained_model_name = pretrained_model_name
        self.preprocessor = _Preprocessor()
        self.trunk = _Trunk()
        self.head = nn.ModuleList([_P(weight=((EMBED,), "ones"), bias=((EMBED,), 0)), nn.Identity(),
                                   _P(weight=((OUT_DIM, EMBED), 0.02))])
        self.postprocessor = nn.ModuleList([nn.Identity(), nn.Module()])
        self.postprocessor[1].register_buffer("log_logit_scale", torch.ones([]) * math.log(20.0))
        self.final_layer_norm = _P(weight=((EMBED,), "ones"), bias=((EMBED,), 0))
        self._packed = None
        if imagebind_checkpoint is not None:
            self.load_imagebind_checkpoint(imagebind_checkpoint)

    def load_imagebind_checkpoint(self, path: str):
        """ImageBind's checkpoint is the state dict of the whole multi-modal model; the audio branch lives under
        `modality_{preprocessors,trunks,heads,postprocessors}.audio.` — the four sub-modules the reference picks out
        (segmask_imagebind.py:58-61).  final_layer_norm is not part of it and keeps its identity initialisation."""
        full = torch.load(path, map_location="cpu", weights_only=True)
        ren = {"modality_preprocessors.audio.": "preprocessor.", "modality_trunks.audio.": "trunk.",
               "modality_heads.audio.": "head.", "modality_postprocessors.audio.": "postprocessor."}
        sd = {}
        for k, v in full.items():
            for old, new in ren.items():
                if k.startswith(old):
                    sd[new + k[len(old):]] = v
        own = self.state_dict()
        missing = [k for k in own if k not in sd and not k.startswith("final_layer_norm.")]
        if missing:
            raise KeyError(f"{path}: audio branch lacks {len(missing)} tensors, e.g. {missing[:3]}")
        sd.update({k: own[k] for k in own if k.startswith("final_layer_norm.")})
        self.load_state_dict({k: sd[k] for k in own})
        return self

    # -- diffusers-style I/O (the trainer saves this module with ModelMixin.save_pretrained) ------------------------
    @classmethod
    def from_config(cls, config: Dict[str, Any], **kw):
        args = {k: v for k, v in dict(config).items() if k in ("n_segment", "pretrained_model_name")}
        args.update(kw)
        return cls(**args)

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, subfolder: Optional[str] = None, **kw):
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(path, CONFIG_NAME)) as f:
            model = cls.from_config(json.load(f), **kw)
        st = os.path.join(path, SAFETENSORS_NAME)
        if os.path.isfile(st):
            from safetensors.torch import load_file

            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, BIN_NAME), map_location="cpu", weights_only=True)
        own = model.state_dict()
        # the checkpoint holds all of ImageBind's audio modules; anything outside the restated subset is reported
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"audio encoder checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
        model.load_state_dict({k: sd[k] for k in own})
        return model.eval()

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True):
        os.makedirs(save_directory, exist_ok=True)
        cfg = {"_class_name": type(self).__name__, "_diffusers_version": "0.29.2"}
        cfg.update(self.config)
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file

            save_file(sd, os.path.join(save_directory, SAFETENSORS_NAME))
        else:
            torch.save(sd, os.path.join(save_directory, BIN_NAME))

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._packed = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = None
        return r

    @property
    def device(self):
        return self.final_layer_norm.weight.device

    @property
    def dtype(self):
        return self.final_layer_norm.weight.dtype

    # -- kernel-side weights ------------------------------------------------------------------------------------
    def pack(self):
        """bf16 GEMM operands, f32 biases / norm parameters / tables, on the module's device."""
        key = (P.ACT, P.SPLIT, str(self.device))
        if self._packed is not None and self._packed.get("key") == key:
            return self._packed
        from .weights import to_act

        bf = lambda t: to_act(t.detach().to(self.device))      # noqa: E731  (split precision: main + rest planes)
        f32 = lambda t: t.detach().to(torch.float32).contiguous()      # noqa: E731
        pre = self.preprocessor
        pk = {
            "stem_w": bf(pre.rgbt_stem.proj.weight.reshape(EMBED, PATCH * PATCH)),
            "stem_g": f32(pre.rgbt_stem.norm_layer.weight), "stem_b": f32(pre.rgbt_stem.norm_layer.bias),
            "cls": f32(pre.cls_token.reshape(EMBED)),
            "pos": f32(pre.pos_embedding_helper.pos_embed.reshape(-1, EMBED)),
            "head_g": f32(self.head[0].weight), "head_b": f32(self.head[0].bias), "head_w": bf(self.head[2].weight),
            "final_g": f32(self.final_layer_norm.weight), "final_b": f32(self.final_layer_norm.bias),
            "blocks": [], "key": key,
        }
        for blk in self.trunk.blocks:
            pk["blocks"].append({
                "n1_g": f32(blk.norm_1.weight), "n1_b": f32(blk.norm_1.bias),
                "in_w": bf(blk.attn.in_proj_weight), "in_b": f32(blk.attn.in_proj_bias),
                "bias_kv": bf(torch.cat([blk.attn.bias_k.reshape(EMBED), blk.attn.bias_v.reshape(EMBED)])),
                "out_w": bf(blk.attn.out_proj.weight), "out_b": f32(blk.attn.out_proj.bias),
                "n2_g": f32(blk.norm_2.weight), "n2_b": f32(blk.norm_2.bias),
                "fc1_w": bf(blk.mlp.fc1.weight), "fc1_b": f32(blk.mlp.fc1.bias),
                "fc2_w": bf(blk.mlp.fc2.weight), "fc2_b": f32(blk.mlp.fc2.bias),
            })
        self._packed = pk
        return pk

    # -- forward ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def trunk_tokens(self, input_features: torch.Tensor) -> torch.Tensor:
        """(b, 1, 128, 204) mel-spectrograms -> trunk output rows, bf16 [b * 230, 768] (row 229 of each clip unused)."""
        if tuple(input_features.shape[1:]) != (1, MEL, FRAMES):
            raise ValueError(f"expected (b, 1, {MEL}, {FRAMES}) mel-spectrograms, got {tuple(input_features.shape)}")
        pk = self.pack()
        b = input_features.shape[0]
        L = 2 + N_FREQ * N_TIME                                    # cls + 228 patches + the bias_kv slot
        x = input_features.to(device=self.device, dtype=torch.float32).contiguous()
        patches = ops.patchify(x, PATCH, PATCH, STRIDE)                                     # [b*228, 256]
        emb = ops.layernorm(ops.gemm(patches, pk["stem_w"]), pk["stem_g"], pk["stem_b"], 1e-5)
        h = ops.vit_tokens(emb, pk["cls"], pk["pos"], b, tail_rows=1)                       # [b*230, 768]
        scale = (EMBED // HEADS) ** -0.5
        for w in pk["blocks"]:
            n1 = ops.layernorm(h, w["n1_g"], w["n1_b"], 1e-6)
            qkv = ops.gemm(n1, w["in_w"], bias=w["in_b"])                                   # [b*230, 2304] = q | k | v
            slot = qkv.view(b, L, 3 * EMBED)[:, L - 1, EMBED:]                              # the appended key/value pair
            slot.copy_(w["bias_kv"].expand_as(slot))
            if P.SPLIT:
                from .weights import rest_of

                rest_of(slot).copy_(rest_of(w["bias_kv"]).expand_as(slot))
            att = ops.attention(qkv[:, :EMBED], qkv[:, EMBED:2 * EMBED], qkv[:, 2 * EMBED:], bq=b, lq=L, lk=L, kv_rows=L,
                                heads=HEADS, q_per_kv=1, frames=1, scale=scale)
            h = ops.gemm(att, w["out_w"], bias=w["out_b"], res1=h)
            n2 = ops.layernorm(h, w["n2_g"], w["n2_b"], 1e-6)
            m = ops.gemm(n2, w["fc1_w"], bias=w["fc1_b"], gelu=True)
            h = ops.gemm(m, w["fc2_w"], bias=w["fc2_b"], res1=h)
        return h

    @torch.no_grad()
    def forward(self, input_features: torch.Tensor, normalize: bool = False, return_dict: Optional[bool] = None):
        if normalize:
            raise NotImplementedError("normalize=True (ImageBind postprocessor) — the pipeline calls normalize=False (:174)")
        pk = self.pack()
        b = input_features.shape[0]
        L = 2 + N_FREQ * N_TIME
        h = self.trunk_tokens(input_features)
        enc = ops.from_act(ops.layernorm(h, pk["final_g"], pk["final_b"], 1e-6).view(b, L, EMBED)[:, :L - 1]).contiguous()   # (b, 229, 768)
        cls_rows = ops.layernorm(h.view(b, L, EMBED)[:, 0], pk["head_g"], pk["head_b"], 1e-6)            # row stride L * EMBED
        cls_embeds = ops.linear_small_m(ops.from_act(cls_rows).contiguous(), pk["head_w"], None)         # (b, 1024)
        masks = audio_segment_mask(self.n_segment).to(enc.device)[None].expand(b, -1, -1).contiguous()
        if not return_dict:
            return cls_embeds, enc, masks
        return ImageBindSegmaskAudioEncoderOutput(audio_embeds=cls_embeds, audio_encodings=enc, audio_segment_masks=masks)

    __call__ = forward
