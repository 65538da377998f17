"""End-to-end clip generation on one MI355X with the reference's schedule (PNDM, num_inference_steps=50 -> 51 UNet
forwards, audio guidance 4.0) through AudioCondAnimationPipeline: SD1.5-shaped UNet + SD1.5 VAE decoder, random
weights, synthetic conditioning.  Reports per-clip latency split into denoising and VAE decode."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from asva_amd.pipeline import AudioCondAnimationPipeline, synthetic_clip
from asva_amd.schedulers import PNDMScheduler
from asva_amd.vae import AutoencoderKL
from asva_amd.conditioning import audio_segment_mask

dev = torch.device("cuda", 0)
unet = bench.build_unet(dev, 0, 1)
with torch.device(dev):
    vae = AutoencoderKL().eval()
pipe = AudioCondAnimationPipeline(unet=unet, scheduler=PNDMScheduler(), vae=vae).to(dev)
pipe.set_progress_bar_config(disable=True)

def run(seed):
    c = synthetic_clip(seed, device=dev)
    kw = dict(texts=[""], text_encodings=[c["text_encodings"][None]], image_latents=c["image_latents"][None], noise=c["noise"][None],
              audio_encodings=c["audio_encodings"][None], null_audio_encodings=c["null_audio_encodings"][None],
              audio_masks=audio_segment_mask(12), num_inference_steps=50, audio_guidance_scale=4.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = pipe(**kw, output_latents=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    vid = pipe.decode_latents(lat.permute(0, 2, 1, 3, 4).reshape(12, 4, 32, 32))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1, vid

run(0)                        # autotune + graph capture
res = [run(s)[:2] for s in (1, 2, 3)]
den = sum(r[0] for r in res) / 3; dec = sum(r[1] for r in res) / 3
# device-only VAE time (decode_latents above includes the 9.4 MB device->host copy of the frames)
z = torch.randn(12, 4, 32, 32, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): vae.decode(z, postprocess=True)
torch.cuda.synchronize(); vdev = (time.perf_counter() - t0) / 5
print(json.dumps({"clip": "12x256x256, PNDM-50 (51 UNet CFG forwards), audio guidance 4.0", "denoise_s": round(den, 4),
                  "decode_incl_d2h_s": round(dec, 4), "vae_decode_device_s": round(vdev, 4), "vae_tflops": round(7.47 / vdev, 1),
                  "clips_per_s": round(1 / (den + dec), 3), "unet_steps_per_s": round(51 / den, 2)}))
