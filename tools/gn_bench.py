"""GPU time of the GroupNorm pair (stats + apply) on the UNet's shapes (graph-replayed)."""
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
for nb, rows, C in [(2, 12 * 1024, 320), (24, 1024, 320), (2, 12 * 1024, 640), (2, 12 * 256, 640), (24, 256, 640), (2, 12 * 64, 1280), (2, 12 * 64, 2560), (2, 12 * 16, 1280)]:
    x = torch.randn(nb * rows, C, device="cuda").bfloat16(); y = torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    t = gtime(lambda: ops.groupnorm(x, None, nb, rows, 32, g, b, 1e-5, True, out=y))
    from asva_amd import _lib
    L = _lib.lib()
    nch = L.avsd_groupnorm_nchunks(nb, rows, C)
    part = torch.empty(L.avsd_groupnorm_scratch_floats(nb, nch, 32, C), device="cuda")
    st = lambda: L.avsd_groupnorm_stats(x.data_ptr(), C, C, None, 0, 0, nb, rows, 32, part.data_ptr(), nch, torch.cuda.current_stream().cuda_stream)
    ap = lambda: L.avsd_groupnorm_apply(x.data_ptr(), C, C, None, 0, 0, nb, rows, 32, g.data_ptr(), b.data_ptr(), 1e-5, part.data_ptr(), nch, 1, y.data_ptr(), C, torch.cuda.current_stream().cuda_stream)
    print(f"nb={nb:2d} rows={rows:6d} C={C:5d}: pair {t:6.1f} us = stats {gtime(st):5.1f} ({nch} chunks) + apply {gtime(ap):5.1f}  ({x.numel() * 2 * 3 / 1e6:.1f} MB moved)")
print("apply time vs nchunks (nb=2, rows=3072, C=640):")
nb, rows, C = 2, 3072, 640
x = torch.randn(nb * rows, C, device="cuda").bfloat16(); y = torch.empty_like(x)
g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
for nch in (8, 16, 32, 64, 128):
    part = torch.empty(L.avsd_groupnorm_scratch_floats(nb, nch, 32, C), device="cuda")
    L.avsd_groupnorm_stats(x.data_ptr(), C, C, None, 0, 0, nb, rows, 32, part.data_ptr(), nch, torch.cuda.current_stream().cuda_stream)
    ap = lambda: L.avsd_groupnorm_apply(x.data_ptr(), C, C, None, 0, 0, nb, rows, 32, g.data_ptr(), b.data_ptr(), 1e-5, part.data_ptr(), nch, 1, y.data_ptr(), C, torch.cuda.current_stream().cuda_stream)
    st = lambda: L.avsd_groupnorm_stats(x.data_ptr(), C, C, None, 0, 0, nb, rows, 32, part.data_ptr(), nch, torch.cuda.current_stream().cuda_stream)
    print(f"  nchunks={nch:4d}: stats {gtime(st):5.1f} us, apply {gtime(ap):5.1f} us")
