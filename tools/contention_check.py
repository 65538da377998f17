"""Correctness while ANOTHER PROCESS keeps the GPU busy (round 4: the asm tiles left fragment prefetches in flight at the end of their
asm block; on an exclusive GPU the data always landed in time, with a second process on the CUs 1-2 % of the launches were wrong).

    python tools/contention_check.py [--reps 1000] [--rounds 8]

Starts `bench.py` in a child process as the noise, then (1) runs every GEMM tile family REPS times on three low-resolution shapes
against an f32 reference and (2) runs the SD1.5-shaped UNet forward ROUNDS times and compares the output hashes."""
import argparse
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from asva_amd import ops  # noqa: E402
from asva_amd.conditioning import audio_segment_mask  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=1000)
    ap.add_argument("--rounds", type=int, default=8)
    a = ap.parse_args()
    noise = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100000", "--warmup", "5", "--no-cpu-baseline", "--no-vae",
                              "--no-roofline", "--no-precise", "--also-clips", "0"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        time.sleep(25)                      # the child builds its model and starts replaying
        dev = torch.device("cuda", 0)
        g = torch.Generator().manual_seed(0)
        total = 0
        for M, N, K in ((1536, 1280, 1280), (3072, 1280, 1280), (1536, 2560, 1280)):
            x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
            w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
            gold = x.float() @ w.float().T
            for tile in (60, 61, 62, 63, 64, 65, 66, 67, 11, 20, 30):
                bad = sum(float((ops.gemm(x, w, out_f32=True, tile=tile) - gold).abs().max()) > 0.05 for _ in range(a.reps))
                total += bad
                print(f"{M}x{N}x{K} tile {tile}: {bad}/{a.reps} launches wrong", flush=True)
        # the A-resident N-streaming tile (70): eight independent waves per workgroup, W straight from memory into MFMA operands
        from asva_amd.weights import pack_frag
        for M, N, K in ((3000, 2560, 320), (1536, 5120, 640)):
            x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
            w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
            wf = pack_frag(w)
            gold = x.float() @ w.float().T
            bad = sum(float((ops.gemm(x, w, out_f32=True, tile=70, w_frag=wf) - gold).abs().max()) > 0.05 for _ in range(a.reps))
            total += bad
            print(f"{M}x{N}x{K} tile 70: {bad}/{a.reps} launches wrong", flush=True)
        unet = bench.build_unet(dev, 0, 1)
        lat, text, audio, null_audio = bench.synthetic_clip(dev, 1000)
        xin = torch.cat([lat, lat])
        t2, a2 = torch.cat([text, text]), torch.cat([null_audio, audio])
        hashes = []
        for _ in range(a.rounds):
            out = unet(xin, 981, t2, a2, audio_attention_mask=audio_segment_mask(12)).sample
            hashes.append(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12])
        print("UNet forward hashes:", sorted(set(hashes)), f"over {a.rounds} rounds", flush=True)
        # the same forward under the per-layer precision plan (round 6: the EPI_REST instantiations, three-pass products writing planes + f32,
        # the split-precision form of the sub-pixel upsample convolution)
        from asva_amd import precision as P
        P.set_plan(True)
        try:
            unet._invalidate()
            ph = []
            for _ in range(max(2, a.rounds // 2)):
                out = unet(xin, 981, t2, a2, audio_attention_mask=audio_segment_mask(12)).sample
                ph.append(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12])
        finally:
            P.set_plan(False)
            unet._invalidate()
        print("UNet forward hashes, precision plan:", sorted(set(ph)), f"over {len(ph)} rounds", flush=True)
        plan_ok = len(set(ph)) == 1
        # the VAE decode of a clip (resident 3x3 convolution tiles, wide-head attention, GroupNorm at 256 x 256)
        from asva_amd.vae import AutoencoderKL
        torch.manual_seed(1)
        with torch.device(dev):
            vae = AutoencoderKL(**bench.SD15_VAE).eval()
        z = torch.randn(12, 4, 32, 32, device=dev)
        vh = [hashlib.sha256(vae.decode(z, postprocess="uint8", return_dict=False)[0].cpu().numpy().tobytes()).hexdigest()[:12] for _ in range(max(2, a.rounds // 4))]
        print("VAE decode hashes:", sorted(set(vh)), f"over {len(vh)} rounds", flush=True)
        ok = total == 0 and len(set(hashes[:a.rounds])) == 1 and plan_ok and len(set(vh)) == 1
        print("CONTENTION CHECK", "OK" if ok else "FAILED", flush=True)
        return 0 if ok else 1
    finally:
        noise.kill()
        noise.wait()


if __name__ == "__main__":
    sys.exit(main())
