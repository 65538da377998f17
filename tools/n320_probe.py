"""The 128 x 320 hand-scheduled tile (id 67) against the tiles the table holds for the N = 320 / 640 / 1280 layers: hot, graph-timed,
with the epilogue each layer runs (residual; temporal mix + residual; folded LayerNorm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)


def row(label, fn, cands):
    cells = []
    for t, sk in cands:
        try:
            fn(t, sk)
            torch.cuda.synchronize()
            us = ops._time_hot(lambda tt, s_: fn(tt, s_), (t, sk), reps=8) * 1e3
            cells.append(f"{t}/{sk}:{us:6.1f}")
        except Exception as e:  # noqa: BLE001
            cells.append(f"{t}/{sk}:  n/a")
    print(f"{label:34s} " + "  ".join(cells), flush=True)


for M, N, K in [(24576, 320, 320), (24576, 320, 640), (24576, 320, 960), (24576, 320, 1280), (12288, 320, 320), (6144, 640, 640), (6144, 640, 2560),
                (1536, 1280, 1280), (1536, 1280, 5120), (24576, 960, 320), (24576, 2560, 320)]:
    a, w, res = rnd(M, K), rnd(N, K, sc=K ** -0.5), rnd(M, N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    cands = [(0, 1), (13, 1), (38, 1), (63, 1), (64, 1), (67, 1)]
    if M * N < 256 * 128 * 128:
        cands += [(67, 2), (67, 3), (67, 4), (67, 5)]
    cands = [(t, s) if t else (0, 0) for t, s in cands]
    row(f"plain+res {M}x{N}x{K}", lambda t, s: ops.gemm(a, w, res1=res, out=out, tile=t, split_k=max(s, 1)) if t else ops.gemm(a, w, res1=res, out=out), cands)
for B, hw, C in [(2, 1024, 320), (1, 1024, 320), (2, 256, 640), (2, 64, 1280), (2, 16, 1280)]:
    Fr = 12
    M = B * Fr * hw
    y, w = rnd(M, C), rnd(C, 3 * C, sc=(3 * C) ** -0.5)
    out = torch.empty_like(y)
    cands = [(0, 0), (13, 1), (63, 1), (64, 1), (67, 1)] + ([(67, 2), (67, 3), (67, 5)] if M * C < 256 * 128 * 128 else [])
    row(f"tmix+res {M}x{C}x{3 * C}", lambda t, s: ops.gemm(y, w, res1=y, out=out, mode=ops.TMIX, tmix=(hw, Fr), tile=t, split_k=max(s, 1)) if t
        else ops.gemm(y, w, res1=y, out=out, mode=ops.TMIX, tmix=(hw, Fr)), cands)
