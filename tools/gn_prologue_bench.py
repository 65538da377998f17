"""GroupNorm + SiLU + 3x3 convolution: the GroupNorm kernels followed by the resident-tile convolution against the convolution
with the GroupNorm prologue (statistics + table + conv).  Hot, graph-timed.  python tools/gn_prologue_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_conv3x3

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
shapes = [(2, 12, 32, 32, 320, 0, 320), (2, 12, 32, 32, 320, 320, 320), (2, 12, 32, 32, 640, 320, 320), (2, 12, 16, 16, 640, 0, 640),
          (2, 12, 16, 16, 640, 640, 640), (2, 12, 16, 16, 1280, 640, 640), (2, 12, 8, 8, 1280, 0, 1280), (2, 12, 8, 8, 1280, 1280, 1280)]
for nb, Fr, hs, ws, c1, c2, cout in shapes:
    rows_b, cin = Fr * hs * ws, c1 + c2
    M = nb * rows_b
    x1 = torch.randn(M, c1, generator=g).to(torch.bfloat16).to(dev)
    x2 = torch.randn(M, c2, generator=g).to(torch.bfloat16).to(dev) if c2 else None
    w = pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).to(torch.bfloat16).to(dev))
    b = torch.randn(cout, generator=g).to(dev)
    gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    conv = (nb * Fr, hs, ws, 1, 0)
    out = torch.empty(M, cout, dtype=torch.bfloat16, device=dev)
    normed = torch.empty(M, cin, dtype=torch.bfloat16, device=dev)
    table = ops.groupnorm_table(x1, x2, nb, rows_b, 32, gamma, beta, 1e-5)
    line = f"nb {nb} {Fr}x{hs}x{ws} c {c1}+{c2} -> {cout}:"
    t_gn = ops._time_hot(lambda *_: ops.groupnorm(x1, x2, nb, rows_b, 32, gamma, beta, 1e-5, True, out=normed), ()) * 1e3
    t_tab = ops._time_hot(lambda *_: ops.groupnorm_table(x1, x2, nb, rows_b, 32, gamma, beta, 1e-5), ()) * 1e3
    line += f" groupnorm {t_gn:6.1f} us, stats+table {t_tab:6.1f} us |"
    for cand in ops.conv3r_candidates(hs, ws, cin, M, cout, gn=(c1, rows_b)):
        if cand[1] not in (1, 4, 5):
            continue
        try:
            t_plain = ops._time_hot(lambda t, sk: ops.gemm(normed, w, bias=b, out=out, mode=ops.CONV3, conv=conv, tile=t, split_k=sk), cand) * 1e3
            t_fused = ops._time_hot(lambda t, sk: ops.gemm(x1, w, a2=x2, bias=b, out=out, mode=ops.CONV3, conv=conv, tile=t, split_k=sk, gn=(table, rows_b)), cand) * 1e3
            line += f"  {cand[0]}/{cand[1]}: {t_plain:6.1f} -> {t_fused:6.1f}"
        except Exception as e:  # noqa: BLE001
            line += f"  {cand}: {str(e)[:40]}"
    print(line, flush=True)
