"""BASELINE.json configs[0] / BASELINE.md section 4 plumbing check, run once per round: a full 25-step DDIM clip at full shape
(SD1.5-shaped 1.17 B-parameter UNet, closed-form filler weights, latents (1, 4, 12, 32, 32), audio guidance 4.0) on the
MI355X path and through the fp32 CPU restatement of the reference loop (oracle/pipeline_ref.py), same latents / noise /
encodings; reports the rel-L2 of the final latents (and of a few intermediate steps) for each storage precision.
~4 minutes of host CPU for the oracle.  Writes JSON to stdout."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from asva_amd import precision as P
from asva_amd.conditioning import audio_segment_mask
from asva_amd.engine import DenoiseEngine
from asva_amd.schedulers import DDIMScheduler
from oracle import pipeline_ref
from oracle.filler import seeded_randn
from tests.helpers import GOLDEN, filled_unet, rel_l2

STEPS = int(os.environ.get("CFG1_STEPS", "25"))
cfg = json.load(open(os.path.join(GOLDEN, "unet_sd15_config.json")))
lat = seeded_randn(61, 1, 4, 12, 32, 32)
lat[:, :, 0] *= 0.18215
text, audio, null_audio = seeded_randn(62, 1, 77, 768), seeded_randn(63, 1, 229, 768), seeded_randn(64, 1, 229, 768)
mask = audio_segment_mask(12)
torch.set_num_threads(min(32, os.cpu_count() or 8))
m = filled_unet(cfg)
sd = {k: v.detach().float() for k, v in m.state_dict().items()}
t0 = time.time()
trace = []
with torch.no_grad():
    ref = pipeline_ref.denoise(sd, dict(m.config), lat, text, audio, null_audio, mask, STEPS, 4.0, "ddim", trace=trace)
t_cpu = time.time() - t0
out = {"workload": f"{STEPS} DDIM steps, CFG 4.0, latents (1,4,12,32,32), SD1.5-shaped UNet (filler weights)",
       "oracle_cpu_seconds": round(t_cpu, 1), "oracle_threads": torch.get_num_threads(), "modes": {}}
for prec, f32 in (("bf16", False), ("bf16", True), ("fp16", False), ("fp16", True)):
    P.set_precision(prec)
    u = filled_unet(cfg).to("cuda")
    u.f32_residual = f32
    eng = DenoiseEngine(u, DDIMScheduler(), audio_guidance_scale=4.0)
    eng.set_conditioning(text.cuda(), audio.cuda(), null_audio.cuda(), mask, 12)
    x = lat.cuda().clone()
    eng.prepare(x, STEPS)
    errs = {}
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(STEPS):
        eng.step(x, i)
        if i + 1 in (1, 5, 10, STEPS):
            errs[str(i + 1)] = rel_l2(x, trace[i])
    torch.cuda.synchronize()
    out["modes"][f"{prec}{'+f32res' if f32 else ''}"] = {"rel_l2_after_steps": errs, "frame0_pinned": bool(torch.equal(x[:, :, 0].cpu(), lat[:, :, 0])),
                                                          "gpu_seconds_incl_checks": round(time.time() - t0, 3)}
    del eng, u
    torch.cuda.empty_cache()
P.set_precision("bf16")
print(json.dumps(out, indent=1))
