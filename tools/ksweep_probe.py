"""Short-K GEMMs: what is fixed per output tile and what grows with K?  M x N fixed, K swept from one 64-wide K tile up;
own tiles against torch.matmul (hipBLASLt), hot, graph-timed, plain 16-bit output.  A straight-line fit gives the fixed
part (launch + prologue + epilogue per round of tiles) and the slope per K tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
KS = (64, 128, 192, 320, 640, 1280)
for M, N, tiles in ((24576, 2560, (11, 14, 63, 61, 60)), (6144, 5120, (11, 14, 63, 61, 60)), (24576, 320, (11, 30, 63, 64, 65, 66)), (6144, 640, (30, 63, 64, 65, 66))):
    print(f"== M={M} N={N}   us per launch at K = {KS}")
    rows = {}
    for K in KS:
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        torch.matmul(a, w.t(), out=out); torch.cuda.synchronize()
        rows.setdefault("lib", []).append(ops._time_hot(lambda tt, sk: torch.matmul(a, w.t(), out=out), (0, 1), reps=8) * 1e3)
        for t in tiles:
            try:
                rows.setdefault(t, []).append(ops._time_hot(lambda tt, sk: ops.gemm(a, w, out=out, tile=tt), (t, 1), reps=8) * 1e3)
            except Exception:  # noqa: BLE001
                rows.setdefault(t, []).append(float("nan"))
    for k, v in rows.items():
        slope = (v[4] - v[3]) / 5.0                      # us per K tile between K = 320 and 640
        fixed = v[3] - 5 * slope
        print(f"  {str(k):>4}: " + " ".join(f"{x:7.1f}" for x in v) + f"    fixed {fixed:6.1f} us + {slope:5.2f} us per K tile")
