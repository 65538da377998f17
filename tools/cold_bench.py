"""What a GEMM launch pays for cold operands: the same launch timed back to back with (a) everything re-used (hot L2),
(b) a different copy of the weights each launch, (c) a different copy of the activations each launch, (d) both.  The
copies of one operand total > 600 MB, more than L2 + Infinity Cache, so a rotating operand always comes from HBM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

dev = "cuda"
SHAPES = [("plain+res", 24576, 320, 320), ("plain", 1536, 1280, 1280), ("plain+res", 6144, 640, 640), ("plain", 24576, 320, 1280),
          ("plain", 1536, 1280, 5120), ("plain", 384, 1280, 1280)]
for kind, M, N, K in SHAPES:
    na = max(2, (600 << 20) // (M * K * 2))
    nw = max(2, (600 << 20) // (N * K * 2))
    na, nw = min(na, 256), min(nw, 256)
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(na)]
    W = [(0.05 * torch.randn(N, K, device=dev)).bfloat16() for _ in range(nw)]
    R = [torch.randn(M, N, device=dev).bfloat16() for _ in range(na)] if "res" in kind else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(A[0], W[0], res1=R[0] if R else None, out=out)          # autotune (cold mode)
    res = {}
    for mode in ("hot", "coldW", "coldA", "coldAW"):
        def run(i):
            a = A[i % na] if "A" in mode else A[0]
            w = W[i % nw] if "W" in mode else W[0]
            r = (R[i % na] if "A" in mode else R[0]) if R else None
            ops.gemm(a, w, res1=r, out=out)
        n = max(na, nw) * 2
        for i in range(8): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): run(i)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * M * N * K
    print(f"{kind:10s} {M:6d} {N:5d} {K:5d} tile={ops._TILE_CACHE and list(ops._TILE_CACHE.values())[-1]}  " +
          "  ".join(f"{k}: {v:6.1f} us ({fl / v / 1e6:5.0f} TF)" for k, v in res.items()))
    del A, W, R
    torch.cuda.empty_cache()
