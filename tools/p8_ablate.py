"""Times tile 37 (csrc/gemm8p.hip) at 4096^3 and 8192^3 with whatever library AVSD_LIB_PATH points to — used with builds of
gemm8p.hip compiled with -DG8_ABL=n (profiles/r3_8phase_probe.txt lists the variants)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from asva_amd import ops  # noqa: E402
from tools.p8_probe import gtime  # noqa: E402

for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (0.02 * torch.randn(N, K, device="cuda")).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = gtime(lambda: ops.gemm(a, w, out=out, tile=ops.TILE_8PHASE))
    print(os.environ.get("AVSD_LIB_PATH", "shipped library")[-16:], M, f"{t:8.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF", flush=True)
