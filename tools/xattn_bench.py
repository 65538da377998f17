"""Micro-benchmark of avsd_cross_attention_block at the UNet's level-0 shape (M = 2 x 12 x 1024 rows, C = 320, 8 heads):
audio (25 gathered keys per frame) and text (77 keys) variants, against the three kernels it replaces.  Times come from a
captured graph of `reps` back-to-back launches.  Under rocprofv3 --pmc use --reps 3 --no-graph."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from asva_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--no-graph", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, Fr, L, C, heads = 2, 12, 1024, 320, 8
M = B * Fr * L
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
rndf = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
stats = torch.empty(M, C // 32, 2, device=dev)
h = ops.gemm(rnd(M, C), rnd(C, C, sc=C ** -0.5), res1=rnd(M, C), rowstats=stats)
wq, wo, bo = rnd(C, C, sc=C ** -0.5), rnd(C, C, sc=C ** -0.5), rndf(C)
colsum, qb = wq.float().sum(1), rndf(C, sc=0.1)
stats_out = torch.empty_like(stats)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    if a.no_graph:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return float("nan")
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    gr.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3


for kind, lk, per_frame in (("audio", 25, True), ("text", 77, False)):
    nkv = B * Fr if per_frame else B
    lkp = (lk + 31) // 32 * 32
    kk, vv = rnd(nkv, lk, C), rnd(nkv, lk, C)
    k_pad = torch.zeros(nkv, lkp, C, dtype=torch.bfloat16, device=dev)
    vt_pad = torch.zeros(nkv, C, lkp, dtype=torch.bfloat16, device=dev)
    k_pad[:, :lk] = kk
    vt_pad[:, :, :lk] = vv.transpose(1, 2)
    qpk = 1 if per_frame else Fr
    out = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    fused = lambda: ops.cross_attention_block(h, stats, wq, colsum, qb, k_pad, vt_pad, lk, wo, bo, res=h, heads=heads, L=L,
                                              q_per_kv=qpk, rowstats=stats_out, out=out)
    q = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    o = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    k2, v2 = kk.reshape(nkv * lk, C), vv.reshape(nkv * lk, C)

    def sep():
        ops.gemm(h, wq, bias=qb, ln=(stats, colsum, 1e-5), out=q)
        ops.attention(q, k2, v2, bq=B * Fr, lq=L, lk=lk, kv_rows=lk, heads=heads, q_per_kv=qpk, frames=Fr, out=o)
        ops.gemm(o, wo, bias=bo, res1=h, rowstats=stats_out, out=out)

    if os.environ.get("AVSD_XATTN_TIMING"):      # a library built with per-phase cycle stamps (xattn_dbg, see tools/README.md): phase split per workgroup
        import ctypes, numpy as np
        from asva_amd import _lib
        for _ in range(3):
            fused()
        torch.cuda.synchronize()
        buf = np.zeros(512 * 8, dtype=np.uint64)
        rc = _lib.lib().avsd_xattn_debug_read(ctypes.c_void_p(buf.ctypes.data))
        d = buf.reshape(512, 8)[: M // 128].astype(np.float64)
        ph = np.diff(d[:, :6], axis=1)
        names = ("stage 1 (Q = LN-fold(h) Wq)", "Q conversion + K / V^T staging", "stage 2 (attention, 8 heads)", "stage 3 (O Wo)", "epilogue")
        print(f"{kind}: cycles per phase, median over {M // 128} workgroups (100 MHz s_memtime ticks x ?): rc {rc}")
        for n_, col in zip(names, ph.T):
            print(f"   {n_:34s} median {np.median(col):9.0f}  min {col.min():9.0f}  max {col.max():9.0f}")
        print(f"   whole workgroup                    median {np.median(d[:, 5] - d[:, 0]):9.0f};  first start -> last end {d[:, 5].max() - d[:, 0].min():9.0f}")
    t_f, t_s = timed(fused, a.reps), timed(sep, a.reps)
    fl = 4.0 * M * C * C + 4.0 * M * lk * C
    print(f"{kind:6s} lk={lk:3d}: fused {t_f:7.1f} us ({fl / t_f / 1e6:6.1f} TFLOP/s)   separate {t_s:7.1f} us")
