"""GPU time of GroupNorm on the UNet's shapes (graph-replayed, hot L2): the stats + apply pair against the one-launch form
(csrc/groupnorm_fused.hip).  profiles/r3_gn_probe.txt is this script's output."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from asva_amd import _lib, ops  # noqa: E402


def gtime(fn, reps=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


L = _lib.lib()
# (nb, rows per batch, c1, c2, launches of this shape in one cfg-2 step)
SHAPES = [(2, 12288, 320, 0, 5), (2, 12288, 320, 320, 2), (2, 12288, 640, 320, 1), (24, 1024, 320, 0, 5),
          (2, 3072, 320, 0, 1), (2, 3072, 640, 0, 7), (2, 3072, 640, 640, 1), (2, 3072, 1280, 640, 1), (2, 3072, 640, 320, 1), (24, 256, 640, 0, 5),
          (2, 768, 640, 0, 1), (2, 768, 1280, 0, 7), (2, 768, 1280, 1280, 2), (2, 768, 1280, 640, 1), (24, 64, 1280, 0, 5),
          (2, 192, 1280, 0, 9), (2, 192, 1280, 1280, 3), (24, 16, 1280, 0, 1)]
tot_pair = tot_best = 0.0
for nb, rows, c1, c2, count in SHAPES:
    C = c1 + c2
    x1 = torch.randn(nb * rows, c1, device="cuda").bfloat16()
    x2 = torch.randn(nb * rows, c2, device="cuda").bfloat16() if c2 else None
    y = torch.empty(nb * rows, C, device="cuda", dtype=torch.bfloat16)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ops._GN_FUSED = False
    pair = gtime(lambda: ops.groupnorm(x1, x2, nb, rows, 32, g, b, 1e-5, True, out=y))
    ok = L.avsd_groupnorm_fused_supported(nb, rows, 32, c1, c2, 0)
    if ok:
        ops._GN_FUSED = True
        one = gtime(lambda: ops.groupnorm(x1, x2, nb, rows, 32, g, b, 1e-5, True, out=y))
    tot_pair += pair * count
    tot_best += (min(pair, one) if ok else pair) * count
    print(f"nb={nb:2d} rows={rows:6d} C={c1:4d}+{c2:<4d} x{count}: stats + apply {pair:6.1f} us   one launch " +
          (f"{one:6.1f} us ({pair / one:.2f}x)" if ok else "   (does not fit)") + f"   {(x1.numel() + (x2.numel() if c2 else 0)) * 4 / 1e6:.1f} MB in + out")
print(f"per step, hot: pair everywhere {tot_pair / 1e3:.3f} ms; one launch where it fits {tot_best / 1e3:.3f} ms")
