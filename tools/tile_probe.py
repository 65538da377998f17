"""Per-tile timing of plain GEMMs at chosen (M, N, K): python tools/tile_probe.py M,N,K[,split] ... [--tiles 6,9,18,20,32]
Hot, graph-timed (ops._time_hot), 16-bit output with a bias."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

dev = torch.device("cuda", 0)
tiles = [6, 9, 14, 18, 19, 20, 29, 30, 31, 32]
shapes = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--tiles":
        tiles = [int(t) for t in args.pop(0).split(",")]
    else:
        shapes.append(tuple(int(v) for v in a.split(",")))
shapes = shapes or [(4096, 4096, 4096), (8192, 8192, 8192), (24576, 2560, 320), (6144, 5120, 640), (6144, 1280, 11520), (24576, 640, 5760),
                    (1536, 1280, 11520, 4), (1536, 1280, 11520, 8), (1536, 10240, 1280)]
g = torch.Generator().manual_seed(0)
for sh in shapes:
    M, N, K = sh[:3]
    sk = sh[3] if len(sh) > 3 else 1
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    print(f"== M={M} N={N} K={K} split_k={sk} ({2.0 * M * N * K / 1e9:.1f} GFLOP)")
    for t in tiles:
        try:
            us = ops._time_hot(lambda tt, s: ops.gemm(a, w, bias=b, out=out, tile=tt, split_k=s), (t, sk), reps=8) * 1e3
            print(f"  tile {t:2d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s")
        except Exception as e:  # noqa: BLE001
            print(f"  tile {t:2d}: n/a ({str(e)[:60]})")
