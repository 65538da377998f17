"""Does it pay to have a layer's weights in the Infinity Cache (MALL) when its GEMM starts?  Times plain GEMMs of the low-resolution
levels in three cache states: cold (384 MB written since the weights were last touched, activations re-read), MALL-warm (the same
flush, then activations AND weights re-read by another kernel: L2 contents do not survive a kernel boundary, the memory-side cache
does), hot (back-to-back replays).  Decides whether a next-layer weight prefetch is worth building."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

for (M, N, K, sk) in [(1536, 1280, 1280, 1), (1536, 1280, 3840, 2), (384, 1280, 1280, 1), (384, 1280, 3840, 8), (1536, 1280, 5120, 4), (384, 1280, 11520, 8), (6144, 640, 640, 1)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    res = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, res1=res, out=out)
    key = [k for k in ops.tile_cache() if k[:4] == (0, M, N, K)]
    launch = lambda t, s: ops.gemm(a, w, res1=res, out=out, tile=t, split_k=s)
    from asva_amd.ops import _heuristic_tile
    cand = ops.tile_cache().get(key[0]) if key else _heuristic_tile(M, N, K, False, True)
    cold = ops._time_cold(launch, cand, (a, res), reps=9)
    warm = ops._time_cold(launch, cand, (a, res, w), reps=9)
    hot = ops._time_hot(launch, cand)
    print(f"{M:5d} x {N:5d} x {K:5d} tile {cand}: cold {cold * 1e3:6.1f} us   MALL-warm weights {warm * 1e3:6.1f} us   hot {hot * 1e3:6.1f} us   (weights {N * K * 2 / 1e6:.1f} MB)")
