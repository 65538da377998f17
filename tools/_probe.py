import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for M, N, K in [(8192, 8192, 8192), (4096, 4096, 4096), (6144, 5120, 640), (24576, 2560, 320), (98304, 960, 320), (61440, 512, 512), (24576, 1280, 1280), (49152, 1280, 640)]:
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    torch.matmul(a, w.t(), out=out); torch.cuda.synchronize()
    row = [f"{M}x{N}x{K}"]
    us = ops._time_hot(lambda tt, sk: torch.matmul(a, w.t(), out=out), (0, 1), reps=4) * 1e3
    row.append(f"lib {us:7.1f} {2.0 * M * N * K / us / 1e6:5.0f}TF")
    for t in (60, 61, 63):
        for G in (1, 2, 4, 8, 16):
            ops._RASTER_G = G
            us = min(ops._time_hot(lambda tt, sk: ops.gemm(a, w, out=out, tile=tt), (t, 1), reps=4) for _ in range(2)) * 1e3
            row.append(f"t{t}/G{G} {us:7.1f}")
    ops._RASTER_G = 0
    print("  ".join(row), flush=True)
