"""The denoising step in split precision (asva_amd.precision.set_split) on the bench workload: steps/s, for rocprofv3 runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from asva_amd import precision as P
from asva_amd.conditioning import audio_segment_mask
from asva_amd.engine import DenoiseEngine
from asva_amd.schedulers import DDIMScheduler

if os.environ.get("MODE", "split") == "plan":      # MODE=plan: the per-layer precision plan instead (asva_amd/precision_plan.json)
    P.set_plan(True)
else:
    P.set_split(True)
dev = torch.device("cuda", 0)
unet = bench.build_unet(dev, 0, 1)
lat, text, audio, null_audio = bench.synthetic_clip(dev, 1000)
eng = DenoiseEngine(unet, DDIMScheduler(), audio_guidance_scale=4.0)
eng.set_conditioning(text, audio, null_audio, audio_segment_mask(12), 12)
x = lat.clone()
eng.prepare(x, 50)
steps = int(os.environ.get("STEPS", "30"))
for i in range(3):
    eng.step(x, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    eng.step(x, (3 + i) % 50)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{os.environ.get('MODE', 'split')}-precision step: {dt * 1e3:.3f} ms ({1 / dt:.2f} steps/s), finite {bool(torch.isfinite(x).all())}")
