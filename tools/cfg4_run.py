"""BASELINE configs[3] ("Landscapes stretch": 24 frames x 512x512 -> latents 24x64x64) run functionally on one MI355X:
one CFG UNet forward (spatial attention over L = 4096, temporal attention over 24 frames, 13 audio keys per frame) and
the VAE decode of 24 x 512 x 512 frames, random weights; reports times and checks finiteness."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from asva_amd.conditioning import audio_segment_mask
from asva_amd.engine import DenoiseEngine
from asva_amd.schedulers import DDIMScheduler
from asva_amd.vae import AutoencoderKL

dev = torch.device("cuda", 0)
unet = bench.build_unet(dev, 0, 1)
g = torch.Generator().manual_seed(0)
lat = torch.randn(1, 4, 24, 64, 64, generator=g).to(dev)
text, audio, null = (torch.randn(1, n, 768, generator=g).to(dev) for n in (77, 229, 229))
eng = DenoiseEngine(unet, DDIMScheduler(), 4.0)
eng.set_conditioning(text, audio, null, audio_segment_mask(24), 24)
eng.prepare(lat, 50)
for i in range(3): eng.step(lat, i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3, 13): eng.step(lat, i)
torch.cuda.synchronize(); step = (time.perf_counter() - t0) / 10
ok1 = bool(torch.isfinite(lat).all())
with torch.device(dev):
    vae = AutoencoderKL().eval()
z = torch.randn(24, 4, 64, 64, generator=g).to(dev)
out = vae.decode(z, postprocess=True).sample
torch.cuda.synchronize(); t0 = time.perf_counter()
out = vae.decode(z, postprocess=True).sample
torch.cuda.synchronize(); dec = time.perf_counter() - t0
print(json.dumps({"config": "24 frames x 512x512 (latents 2x4x24x64x64 with CFG)", "unet_step_ms": round(step * 1e3, 2),
                  "steps_per_s": round(1 / step, 2), "step_tflops": round(47.77 / step, 1), "latents_finite": ok1,
                  "vae_decode_24x512x512_s": round(dec, 3), "vae_tflops": round(60.35 / dec, 1), "frames_finite": bool(torch.isfinite(out).all()),
                  "frames_shape": list(out.shape), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
