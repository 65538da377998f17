"""What would an IN-WORKGROUP split-K be worth?  The asm tiles on the step's 240-workgroup layers with split_k = 1 / 2 / 4, and the reduce launch alone:
the GEMM part of a split run (total minus reduce) bounds what a workgroup whose wave groups walk K slices side by side could reach."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)


def t(fn):
    fn(); torch.cuda.synchronize()
    return ops._time_hot(lambda *_: fn(), ()) * 1e3


for M, N, K, tile in [(1536, 1280, 1280, 65), (6144, 640, 640, 63), (1536, 1280, 5120, 65), (6144, 640, 2560, 63), (384, 1280, 1280, 66), (1536, 1280, 2560, 65)]:
    a, w, res = rnd(M, K), rnd(N, K, sc=K ** -0.5), rnd(M, N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = [f"plain+res {M}x{N}x{K} tile {tile}:"]
    for sk in (1, 2, 4):
        row.append(f"split {sk} {t(lambda: ops.gemm(a, w, res1=res, out=out, tile=tile, split_k=sk)):6.1f} us")
    # cold weights: a fresh copy of W per launch is what the step sees; approximate with 8 rotating copies
    ws = [w.clone() for _ in range(8)]
    i = [0]
    def cold(sk):
        i[0] = (i[0] + 1) % 8
        ops.gemm(a, ws[i[0]], res1=res, out=out, tile=tile, split_k=sk)
    print(" ".join(row), flush=True)
for B, hw, C, tile in [(2, 64, 1280, 65), (2, 256, 640, 65)]:
    Fr = 12
    M = B * Fr * hw
    y, w = rnd(M, C), rnd(C, 3 * C, sc=(3 * C) ** -0.5)
    out = torch.empty_like(y)
    row = [f"tmix+res {M}x{C}x{3 * C} tile {tile}:"]
    for sk in (1, 2, 3):
        row.append(f"split {sk} {t(lambda: ops.gemm(y, w, res1=y, out=out, mode=ops.TMIX, tmix=(hw, Fr), tile=tile, split_k=sk)):6.1f} us")
    print(" ".join(row), flush=True)
