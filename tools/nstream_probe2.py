"""Would the A-resident N-streaming tile (70) pay on the OTHER short-K layers of the 32 x 32 / 16 x 16 levels (square projections, fused q|k|v)?
Hot, graph-timed, with the epilogue each layer runs; explicit tile 70 against the table's pick and torch.matmul."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_frag

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
for M, N, K, kind in [(24576, 320, 320, "res+stats"), (24576, 320, 320, "ln"), (24576, 960, 320, "ln"), (24576, 640, 320, "plain"), (12288, 320, 320, "res+stats"),
                      (6144, 640, 640, "res+stats"), (6144, 640, 640, "ln"), (6144, 1920, 640, "ln"), (6144, 1280, 640, "plain"), (98304, 320, 320, "res+stats"), (98304, 960, 320, "ln")]:
    a, w, res = rnd(M, K), rnd(N, K, sc=K ** -0.5), rnd(M, N)
    wf = pack_frag(w)
    bias = torch.randn(N, generator=g).to(dev)
    st_in = torch.empty(M, K // 32, 2, device=dev)
    h = ops.gemm(a, torch.eye(K, device=dev).to(torch.bfloat16), rowstats=st_in)
    st_out = torch.empty(M, N // 32, 2, device=dev)
    cs = w.float().sum(1)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if kind == "res+stats":
        kw = dict(bias=bias, res1=res, rowstats=st_out, out=out)
        src = a
    elif kind == "ln":
        kw = dict(bias=bias, ln=(st_in, cs, 1e-5), out=out)
        src = h
    else:
        kw = dict(out=out)
        src = a
    ref = ops.gemm(src, w, **kw).clone()
    t_tab = ops._time_hot(lambda *_: ops.gemm(src, w, **kw), ()) * 1e3
    got = ops.gemm(src, w, tile=70, w_frag=wf, **kw)
    torch.cuda.synchronize()
    err = ((got.float() - ref.float()).norm() / ref.float().norm()).item()
    t_ns = ops._time_hot(lambda *_: ops.gemm(src, w, tile=70, w_frag=wf, **kw), ()) * 1e3
    o_l = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    torch.matmul(a, w.t(), out=o_l)
    torch.cuda.synchronize()
    t_lib = ops._time_hot(lambda *_: torch.matmul(a, w.t(), out=o_l), ()) * 1e3
    print(f"{M:6d} x {N:5d} x {K:4d} {kind:10s}: table {t_tab:6.1f} us   nstream {t_ns:6.1f} us   torch.matmul plain {t_lib:6.1f} us   (rel diff vs table {err:.1e})", flush=True)
