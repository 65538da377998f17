"""SD1.5 VAE decode of one 12-frame clip, a few repetitions — run under `rocprofv3 --kernel-trace --stats` for the per-kernel
split of the decoder (profiles/r3_vae_trace.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from asva_amd.vae import AutoencoderKL  # noqa: E402

dev = torch.device("cuda", 0)
with torch.device(dev):
    vae = AutoencoderKL(**bench.SD15_VAE).eval()
z = torch.randn(12, 4, 32, 32, device=dev)
for _ in range(2):
    vae.decode_to_uint8_frames(z.reshape(1, 12, 4, 32, 32).permute(0, 2, 1, 3, 4).contiguous()) if hasattr(vae, "decode_to_uint8_frames") else vae.decode(z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    vae.decode(z, postprocess="uint8")
e1.record()
torch.cuda.synchronize()
print(f"VAE decode: {e0.elapsed_time(e1) / 5:.2f} ms per 12-frame clip")
