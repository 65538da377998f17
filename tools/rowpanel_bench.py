"""The short-K linear layers of the 32 x 32 level: tuned gemm2 tile against the row-panel tile (csrc/rowpanel.hip).  Hot, graph-timed.
python tools/rowpanel_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_geglu

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for M in (24576, 98304, 12288):
    for N, K, kind in ((320, 320, "res"), (320, 320, "ln"), (960, 320, "plain"), (640, 320, "plain"), (2560, 320, "geglu+ln"), (320, 64, "res")):
        a = (torch.randn(M, K, generator=g) * 2 + 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        kw = dict(bias=b)
        n_out = N
        if kind == "res":
            kw["res1"] = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
            kw["rowstats"] = torch.empty(M, N // 32, 2, device=dev)
        if "ln" in kind:
            hb = a.float().reshape(M, K // 32, 32)
            kw["ln"] = (torch.stack([hb.sum(-1), (hb * hb).sum(-1)], -1).contiguous(), w.float().sum(1), 1e-5)
        if "geglu" in kind:
            kw["geglu"] = True
            n_out = N // 2
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=dev)
        res = {}
        for cand in ((4, 1), (6, 1), (11, 1), (12, 1), (13, 1), (14, 1), (15, 1), (17, 1), (19, 1), (20, 1), (23, 1), (32, 1), (33, 1), (50, 1)):
            try:
                ops.gemm(a, w, out=out, tile=cand[0], **kw)
                res[cand[0]] = ops._time_hot(lambda t, sk: ops.gemm(a, w, out=out, tile=t, **kw), cand) * 1e3
            except Exception:  # noqa: BLE001
                pass
        best = min((t for t in res if t != 50), key=res.get)
        fl = 2.0 * M * N * K
        print(f"M {M:6d} N {N:5d} K {K:4d} {kind:9s}: best gemm2 tile {best:2d} {res[best]:7.1f} us {fl / res[best] / 1e6:5.0f} TF | rowpanel {res.get(50, float('nan')):7.1f} us "
              f"{fl / res.get(50, float('nan')) / 1e6:5.0f} TF  x{res[best] / res.get(50, float('nan')):.2f}", flush=True)
