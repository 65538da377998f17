"""Microbenchmark of avsd_attention on the UNet's shapes (time, algorithmic TFLOP/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
dev = "cuda"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
B, F = 2, 12
for name, C, L, Lk, kvrows, qpk in [("spatial32", 320, 1024, 1024, 1024, F), ("spatial16", 640, 256, 256, 256, F), ("spatial8", 1280, 64, 64, 64, F),
                                     ("text32", 320, 1024, 77, 77, F), ("audio32", 320, 1024, 25, 229, F), ("text16", 640, 256, 77, 77, F)]:
    heads = 8
    q = torch.randn(B * F * L, C, device=dev).bfloat16()
    kv = torch.randn(B * kvrows, 2 * C, device=dev).bfloat16()
    idx = None
    if name.startswith("audio"):
        idx = torch.stack([torch.randperm(229)[:25].sort().values for _ in range(F)]).int().to(dev)
    out = torch.empty_like(q)
    dt = timeit(lambda: ops.attention(q, kv[:, :C], kv[:, C:], bq=B * F, lq=L, lk=Lk, kv_rows=kvrows, heads=heads, q_per_kv=qpk, frames=F,
                                      key_index=idx, out=out))
    fl = 4.0 * B * F * heads * L * Lk * (C // heads)
    print(f"{name:10s} d={C//heads:3d} Lq={L:5d} Lk={Lk:5d}  {dt*1e6:8.1f} us  {fl/dt/1e12:7.1f} TFLOP/s")
