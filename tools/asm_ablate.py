import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
names = {60: "full", 64: "no gload", 65: "no dswrite", 66: "no dsread", 67: "no gload+dswrite", 68: "no gload+dswrite+dsread", 69: "no barrier", 70: "mfma only", 20: "v2 tile 20", 61: "asm 256x128", 63: "asm 128x128"}
for M, N, K in [(8192, 8192, 8192), (4096, 4096, 4096), (6144, 5120, 640)]:
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for t, nm in names.items():
        us = min(ops._time_hot(lambda tt, sk: ops.gemm(a, w, out=out, tile=tt), (t, 1), reps=4) for _ in range(2)) * 1e3
        print(f"{M}x{N}x{K} {nm:28s} {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)
