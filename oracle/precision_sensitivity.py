"""Per-layer-group precision sensitivity of the UNet forward, on the CPU oracle (TEST INFRASTRUCTURE, as the rest of oracle/).

Question behind the per-layer precision plan (asva_amd/precision_plan.json): which matrix products of the step may run
ONE MFMA pass on 16-bit main planes (operands rounded to IEEE half, f32 accumulation, un-rounded two-plane output) and
which need the three-pass split product, for the whole forward to stay inside BASELINE.json's 1e-3 rel-L2 of the
reference's fp32 output?

Method: the fp32 oracle forward (oracle/unet_ref.py, pinned to the reference's own UNet, tests/golden) at BASELINE
cfg-2 shape with the closed-form filler weights is the yardstick.  For every group g of products (kind of layer x
resolution level) the forward is repeated with ONLY that group's operands rounded to fp16 at the product's inputs
(weights and the activation operand; the attention groups round q, k, v and the probabilities), everything else
fp32.  e_g = rel-L2 of that output against the fp32 output.  Rounding errors of different products are independent
and small, so a set S of one-pass groups lands at ~ sqrt(sum_{g in S} e_g^2) — checked by the `--verify` run of the
chosen set as a whole.

    python -m oracle.precision_sensitivity --out profiles/r5_precision_sensitivity.json          # ~15 min on 8 cores
    python -m oracle.precision_sensitivity --verify asva_amd/precision_plan.json

Round 6, per OPERAND (`--operand a` / `--operand w`): only the activation operand or only the weight of the group's products is rounded.
A two-pass product (main.main + rest.main, or main.main + main.rest) removes the rounding of ONE operand for two thirds of the three-pass
cost; whether that is enough for a group is read off these rows (profiles/r6_precision_sensitivity_operands.json).  `--verify` plans may
carry "round_a_only" / "round_w_only" lists beside "one_pass" (the operand that stays rounded).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import unet_ref  # noqa: E402

FINE_SAMPLERS = os.environ.get("FINE_SAMPLERS", "0") != "0"
LEVELS = {"down_blocks.0": "r32", "down_blocks.1": "r16", "down_blocks.2": "r8", "down_blocks.3": "r4", "mid_block": "r4",
          "up_blocks.0": "r4", "up_blocks.1": "r8", "up_blocks.2": "r16", "up_blocks.3": "r32"}


def group_of(name: str) -> str | None:
    """state_dict weight name -> precision group 'kind@level' (None: not a matrix product of the step's hot path)"""
    if not name.endswith(".weight"):
        return None
    base = name[: -len(".weight")]
    if base.startswith("time_embedding") or ".pos_embedding_temp." in base or ".time_emb_proj" in base or ".norm" in base or base.startswith("conv_norm_out"):
        return None
    if base in ("conv_in", "conv_in.conv_temp"):
        return "conv_in@r32"
    if base in ("conv_out", "conv_out.conv_temp"):
        return "conv_out@r32"
    blk = re.match(r"((?:down|up)_blocks\.\d|mid_block)\.", base)
    if not blk:
        return None
    lvl = LEVELS[blk.group(1)]
    # up-block samplers feed the next (finer) level; down-block samplers the next coarser: keep them with their block
    rest = base[len(blk.group(0)):]
    if "samplers" in rest:
        # FINE_SAMPLERS: down- and up-samplers apart (the plan can list single samplers: "sampler@up_blocks.2")
        side = ("_down" if rest.startswith("down") else "_up") if FINE_SAMPLERS else ""
        return ("sampler" + side + "_temp@" if rest.endswith("conv_temp") else "sampler" + side + "@") + lvl
    if rest.startswith("resnets"):
        if "conv_shortcut" in rest:
            return ("shortcut_temp@" if rest.endswith("conv_temp") else "shortcut@") + lvl
        if rest.endswith("conv_temp"):
            return "conv_temp@" + lvl
        return "conv3@" + lvl
    if rest.startswith("attentions"):
        for pat, kind in ((".proj_in", "proj_in"), (".proj_out", "proj_out"), (".attn1.to_q", "attn1_q"), (".attn1.to_k", "attn1_kv"),
                          (".attn1.to_v", "attn1_kv"), (".attn1.to_out.0", "attn1_out"), (".attn_audio.to_q", "audio_q"),
                          (".attn_audio.to_k", "audio_kv"), (".attn_audio.to_v", "audio_kv"), (".attn_audio.to_out.0", "audio_out"),
                          (".attn2.to_q", "text_q"), (".attn2.to_k", "text_kv"), (".attn2.to_v", "text_kv"), (".attn2.to_out.0", "text_out"),
                          (".attn_temp.to_q", "temp_qkv"), (".attn_temp.to_k", "temp_qkv"), (".attn_temp.to_v", "temp_qkv"),
                          (".attn_temp.to_out.0", "temp_out"), (".ff.net.0.proj", "ff1"), (".ff.net.2", "ff2")):
            if rest.endswith(pat):
                return kind + "@" + lvl
    return None


class Rounder:
    """patches the oracle's F.linear / F.conv2d / SDPA so that products of the active groups see fp16-rounded operands"""

    def __init__(self, sd):
        self.names = {}
        for k, v in sd.items():
            g = group_of(k)
            if g is not None:
                self.names[id(v)] = (k, g)
        self.active = set()          # groups whose products see rounded operands: a set (both operands) or {group: "both" | "a" | "w"}
        self.sdpa_level = None
        self.count = {}

    def mode(self, g):
        """None (exact), or which operand(s) of group g's products are rounded"""
        if isinstance(self.active, dict):
            return self.active.get(g)
        return "both" if g in self.active else None

    @staticmethod
    def r16(t):
        return t.to(torch.float16).to(torch.float32)

    def groups(self):
        gs = sorted({g for _, g in self.names.values()})
        # attention products have no weight: one group per (which attention, level); the level is taken from the q projection
        gs += [f"{a}_sdpa@{lv}" for a in ("attn1", "audio", "text", "temp") for lv in ("r32", "r16", "r8", "r4")]
        return gs

    def install(self):
        lin, conv, sdpa = F.linear, F.conv2d, F.scaled_dot_product_attention
        me = self

        def linear(x, w, b=None):
            hit = me.names.get(id(w))
            if hit is not None:
                name, g = hit
                if name.endswith("to_q.weight"):          # remember which attention / level the next SDPA belongs to
                    kind = {"attn1_q": "attn1", "audio_q": "audio", "text_q": "text", "temp_qkv": "temp"}[g.split("@")[0]]
                    me.sdpa_level = f"{kind}_sdpa@{g.split('@')[1]}"
                md = me.mode(g)
                if md is not None:
                    me.count[g] = me.count.get(g, 0) + 1
                    return lin(me.r16(x) if md != "w" else x, me.r16(w) if md != "a" else w, b)
            return lin(x, w, b)

        def conv2d(x, w, b=None, **kw):
            hit = me.names.get(id(w))
            md = None if hit is None else me.mode(hit[1])
            if md is not None:
                me.count[hit[1]] = me.count.get(hit[1], 0) + 1
                return conv(me.r16(x) if md != "w" else x, me.r16(w) if md != "a" else w, b, **kw)
            return conv(x, w, b, **kw)

        def attention(q, k, v, attn_mask=None):
            g = me.sdpa_level
            if me.mode(g) is not None:
                me.count[g] = me.count.get(g, 0) + 1
                q, k, v = me.r16(q), me.r16(k), me.r16(v)
                s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
                if attn_mask is not None:
                    s = s.masked_fill(~attn_mask, float("-inf"))
                p = torch.softmax(s, dim=-1)
                return me.r16(p) @ v
            return sdpa(q, k, v, attn_mask=attn_mask)

        unet_ref.F = type("FPatched", (), {})()
        for n in dir(F):
            if not n.startswith("__"):
                setattr(unet_ref.F, n, getattr(F, n))
        unet_ref.F.linear, unet_ref.F.conv2d, unet_ref.F.scaled_dot_product_attention = linear, conv2d, attention


def build_inputs():
    """the model, inputs and yardstick of bench.py's precise_rel_l2 / tests/golden/unet_sd15_forward.pt"""
    from asva_amd.conditioning import audio_segment_mask
    from asva_amd.filler import fill_module_, seeded_randn
    from asva_amd.unet import AudioUNet3DConditionModel

    gdir = os.path.join(ROOT, "tests", "golden")
    g = torch.load(os.path.join(gdir, "unet_sd15_forward.pt"), map_location="cpu", weights_only=True)
    with open(os.path.join(gdir, "unet_sd15_config.json")) as f:
        cfg = json.load(f)
    m = AudioUNet3DConditionModel.from_config(cfg).eval()
    fill_module_(m)
    sd = {k: v.detach().float() for k, v in m.state_dict().items()}
    del m
    lat = seeded_randn(1, 1, 4, 12, 32, 32)
    x = torch.cat([lat, lat])
    text = seeded_randn(2, 1, 77, 768).expand(2, 77, 768)[:, None].expand(2, 12, 77, 768)
    audio = torch.cat([seeded_randn(4, 1, 229, 768), seeded_randn(3, 1, 229, 768)])[:, None].expand(2, 12, 229, 768)
    mask = audio_segment_mask(12)[None].expand(2, -1, -1)
    return sd, cfg, (x, g["timestep"], text, audio, mask), g["full32"].float()


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--verify", default="", help="a plan JSON ({'one_pass': [groups]}): run the whole set at once")
    ap.add_argument("--threads", type=int, default=max(1, (os.cpu_count() or 2) - 2))
    ap.add_argument("--only", default="", help="comma-separated group names (default: all)")
    ap.add_argument("--operand", default="both", choices=("both", "a", "w"), help="which operand of the group's products is rounded")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    sd, cfg, inp, golden = build_inputs()
    rd = Rounder(sd)
    rd.install()
    with torch.no_grad():
        t0 = time.time()
        ref = unet_ref.unet_forward(sd, cfg, *inp)
        print(f"fp32 oracle forward {time.time() - t0:.1f} s; vs the reference's golden fp32 output {rel(ref, golden):.2e}", flush=True)
        if a.verify:
            with open(a.verify) as f:
                plan = json.load(f)
            rd.active = {g: "both" for g in plan["one_pass"]}
            rd.active.update({g: "a" for g in plan.get("round_a_only", ())})
            rd.active.update({g: "w" for g in plan.get("round_w_only", ())})
            out = unet_ref.unet_forward(sd, cfg, *inp)
            print(json.dumps({"one_pass_groups": len(rd.active), "rel_l2_vs_fp32_oracle": rel(out, ref),
                              "rel_l2_vs_reference_golden": rel(out, golden)}))
            return
        groups = rd.groups() if not a.only else a.only.split(",")
        res = {}
        for gname in groups:
            rd.active, rd.count = {gname: a.operand}, {}
            t0 = time.time()
            out = unet_ref.unet_forward(sd, cfg, *inp)
            if not rd.count:
                continue                     # the model has no product in this group
            res[gname] = {"rel_l2": rel(out, ref), "products": rd.count.get(gname, 0)}
            print(f"{gname:18s} {res[gname]['rel_l2']:.3e}  ({res[gname]['products']} products, {time.time() - t0:.0f} s)", flush=True)
        rd.active = {g: a.operand for g in res}
        out = unet_ref.unet_forward(sd, cfg, *inp)
        tot = rel(out, ref)
        quad = sum(v["rel_l2"] ** 2 for v in res.values()) ** 0.5
        print(f"all groups one-pass: {tot:.3e}; sqrt(sum e_g^2) = {quad:.3e}")
        if a.out:
            with open(a.out, "w") as f:
                json.dump({"what": "rel-L2 of the cfg-2 UNet forward vs the fp32 oracle when ONLY this group's products see fp16-rounded operands "
                                   "(oracle/precision_sensitivity.py)", "operand": a.operand, "groups": res, "all_groups_one_pass": tot, "quadrature_sum": quad}, f, indent=1)


if __name__ == "__main__":
    main()
