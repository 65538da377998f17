"""LayerNorm -> Linear as two kernels vs folded into the GEMMs (producer emits row statistics, consumer folds them);
GPU time from captured graphs of 20 repetitions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

def gtime(fn, reps=20):
    for _ in range(2): fn()
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
for M, C in [(24576, 320), (6144, 640), (1536, 1280)]:
    o = torch.randn(M, C, device=dev).bfloat16(); res = torch.randn(M, C, device=dev).bfloat16()
    wo = (0.05 * torch.randn(C, C, device=dev)).bfloat16(); bo = torch.randn(C, device=dev)
    wq = (0.05 * torch.randn(C, C, device=dev)).bfloat16()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.empty(M, C // 32, 2, device=dev)
    cs = wq.float().sum(1); bq = torch.zeros(C, device=dev)
    h = torch.empty(M, C, device=dev, dtype=torch.bfloat16); q = torch.empty_like(h); n = torch.empty_like(h)
    t_prod = gtime(lambda: ops.gemm(o, wo, bias=bo, res1=res, out=h))
    t_prod_s = gtime(lambda: ops.gemm(o, wo, bias=bo, res1=res, out=h, rowstats=stats))
    t_ln = gtime(lambda: ops.layernorm(h, g, b, out=n))
    t_q = gtime(lambda: ops.gemm(n, wq, out=q))
    t_qf = gtime(lambda: ops.gemm(h, wq, bias=bq, out=q, ln=(stats, cs, 1e-5)))
    print(f"M={M} C={C}: out-proj {t_prod:.1f} us, +rowstats {t_prod_s:.1f} | layernorm {t_ln:.1f} + to_q {t_q:.1f} = {t_ln + t_q:.1f} | fused to_q {t_qf:.1f} "
          f"| pair: {t_prod + t_ln + t_q:.1f} -> {t_prod_s + t_qf:.1f} us")
