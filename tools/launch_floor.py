"""Per-kernel cost inside a captured graph: 200 dependent launches of a trivial kernel (timestep embedding of one value),
of a 1-block GEMM and of a 15.7 MB LayerNorm, replayed; reports us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
dev = "cuda"
def gtime(fn, n=200):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
t = torch.tensor([501.0], device=dev)
print(f"timestep_embedding(1 value): {gtime(lambda: ops.timestep_embedding(t, 320)):.2f} us per launch")
a = torch.randn(64, 64, device=dev).bfloat16(); w = torch.randn(64, 64, device=dev).bfloat16(); o = torch.empty(64, 64, device=dev, dtype=torch.bfloat16)
print(f"64x64x64 GEMM (1 block): {gtime(lambda: ops.gemm(a, w, out=o, tile=13)):.2f} us per launch")
x = torch.randn(24576, 320, device=dev).bfloat16(); y = torch.empty_like(x); g = torch.ones(320, device=dev); b = torch.zeros(320, device=dev)
print(f"LayerNorm 24576x320 (31 MB moved): {gtime(lambda: ops.layernorm(x, g, b, out=y)):.2f} us per launch")
x2 = torch.randn(1536, 1280, device=dev).bfloat16(); y2 = torch.empty_like(x2); g2 = torch.ones(1280, device=dev); b2 = torch.zeros(1280, device=dev)
print(f"LayerNorm 1536x1280 (7.9 MB moved): {gtime(lambda: ops.layernorm(x2, g2, b2, out=y2)):.2f} us per launch")
