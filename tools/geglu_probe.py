"""Where does the GEGLU projection (24576 x 2560 x 320, the most expensive GEMM shape of the step) spend its time?
Variants x tiles, hot, graph-timed: plain 16-bit output of all 2560 columns / GEGLU epilogue / GEGLU + folded LayerNorm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_geglu

dev = torch.device("cuda", 0)
shapes = [(24576, 320), (6144, 640), (1536, 1280)]
g = torch.Generator().manual_seed(0)
for M, C in shapes:
    x = (torch.randn(M, C, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(8 * C, C, generator=g) * C ** -0.5).to(dev)
    b = torch.randn(8 * C, generator=g).to(dev)
    wp, bp = pack_geglu(w, b)
    stats = torch.empty(M, C // 32, 2, device=dev)
    h = ops.gemm(x, (torch.eye(C, device=dev)).to(torch.bfloat16), rowstats=stats)
    cs = wp.float().sum(1)
    out_g = torch.empty(M, 4 * C, dtype=torch.bfloat16, device=dev)
    out_p = torch.empty(M, 8 * C, dtype=torch.bfloat16, device=dev)
    variants = {"plain": lambda t: ops.gemm(h, wp, bias=bp, out=out_p, tile=t),
                "geglu": lambda t: ops.gemm(h, wp, bias=bp, geglu=True, out=out_g, tile=t),
                "geglu+ln": lambda t: ops.gemm(h, wp, bias=bp, geglu=True, ln=(stats, cs, 1e-5), out=out_g, tile=t),
                "geglu+ln(prefolded)": lambda t: ops.gemm(h, wp, bias=bp, geglu=True, ln=(folded, cs, 1e-5), out=out_g, tile=t)}
    folded = ops.ln_fold(stats)
    print(f"   ln_fold kernel: {ops._time_hot(lambda *_: ops.ln_fold(stats), ()) * 1e3:.1f} us")
    a_l = torch.randn(M, C, device=dev).bfloat16(); w_l = torch.randn(8 * C, C, device=dev).bfloat16()
    torch.matmul(a_l, w_l.t(), out=out_p); torch.cuda.synchronize()
    t_lib = ops._time_hot(lambda tt, sk: torch.matmul(a_l, w_l.t(), out=out_p), (0, 1), reps=8) * 1e3
    print(f"== M={M} N={8 * C} K={C}  ({2.0 * M * 8 * C * C / 1e9:.1f} GFLOP)   torch.matmul (all {8 * C} columns, no epilogue): {t_lib:.1f} us")
    for t in (11, 14, 9, 61, 62, 63):
        row = []
        for name, fn in variants.items():
            try:
                us = ops._time_hot(lambda tt, sk: fn(tt), (t, 1), reps=8) * 1e3
                row.append(f"{name} {us:7.1f} us")
            except Exception as e:  # noqa: BLE001
                row.append(f"{name} n/a")
        print(f"  tile {t:2d}: " + "   ".join(row))
