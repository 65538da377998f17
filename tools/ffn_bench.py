"""Fused GEGLU feed-forward block (avsd_ffn_block) against the two GEMMs it replaces, graph-timed, at the SD1.5 level-0 shape
(M = 24576 rows per clip with CFG batch 2, C = 320, hidden 1280)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_geglu, pack_linear

def gtime(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3

dev = "cuda"
for M in (24576, 12288, 98304):
    C, NH = 320, 1280
    x = torch.randn(M, C, device=dev).bfloat16()
    w0 = (torch.randn(C, C, device=dev) * C ** -0.5).bfloat16()
    stats = torch.empty(M, C // 32, 2, device=dev)
    h = ops.gemm(x, w0, rowstats=stats)
    w1 = torch.randn(2 * NH, C, device=dev) * C ** -0.5
    b1 = torch.randn(2 * NH, device=dev) * 0.1
    w1p, b1p = pack_geglu(w1, b1)
    s1 = w1p.float().sum(1)
    w2 = pack_linear(torch.randn(C, NH, device=dev) * NH ** -0.5)
    w2c = w2.reshape(C, NH // 16, 16).permute(1, 0, 2).contiguous()
    b2 = torch.randn(C, device=dev) * 0.1
    out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    g = torch.empty(M, NH, device=dev, dtype=torch.bfloat16)
    cb1 = ops.ffn_fold_terms(s1, b1p)
    t_f = gtime(lambda: ops.ffn_block(h, stats, w1p, cb1, w2c, b2, res=h, out=out))
    t_1 = gtime(lambda: ops.gemm(h, w1p, bias=b1p, geglu=True, ln=(stats, s1, 1e-5), out=g))
    t_2 = gtime(lambda: ops.gemm(g, w2, bias=b2, res1=h, out=out))
    fl = 2.0 * M * C * 3 * NH
    print(f"M={M}: fused {t_f:7.1f} us ({fl / t_f / 1e6:5.0f} TF)   FF1 {t_1:7.1f} + FF2 {t_2:7.1f} = {t_1 + t_2:7.1f} us ({fl / (t_1 + t_2) / 1e6:5.0f} TF)   ratio {t_f / (t_1 + t_2):.2f}")
