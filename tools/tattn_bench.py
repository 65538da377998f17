"""GPU time of avsd_temporal_attention on the UNet's shapes (graph-replayed)."""
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
for hw, C in [(1024, 320), (256, 640), (64, 1280), (16, 1280)]:
    qkv = torch.randn(2 * 12 * hw, 3 * C, device="cuda").bfloat16()
    out = torch.empty(2 * 12 * hw, C, device="cuda", dtype=torch.bfloat16)
    t = gtime(lambda: ops.temporal_attention(qkv, b=2, frames=12, hw=hw, heads=8, out=out))
    mb = (qkv.numel() + out.numel()) * 2 / 1e6
    print(f"hw={hw:5d} C={C:5d}: {t:6.1f} us  {mb / t:6.2f} TB/s ({mb:.1f} MB)")
