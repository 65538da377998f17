"""Microbenchmark of avsd_gemm_bf16 on the UNet's real shapes: TFLOP/s per (shape, mode, tile).
    python tools/gemm_bench.py [--quick]
"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_conv3x3

dev = torch.device("cuda")

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def bench_plain(M, N, K, tiles=(0, 4, 6, 9, 13, 17, 18, 20, 23, 24, 25, 26), **kw):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    out = torch.empty(M, N if not kw.get("geglu") else N // 2, device=dev, dtype=torch.bfloat16)
    r = {}
    for t in tiles:
        dt = timeit(lambda: ops.gemm(a, w, out=out, tile=t, **kw))
        r[t] = 2.0 * M * N * K / dt / 1e12
    return r

def bench_conv(n_img, h, w_, cin, cout, tiles=(0, 4, 6, 9, 13, 17, 18, 20, 23, 24, 25, 26)):
    x = torch.randn(n_img * h * w_, cin, device=dev).bfloat16()
    w = pack_conv3x3((torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5))
    out = torch.empty(n_img * h * w_, cout, device=dev, dtype=torch.bfloat16)
    r = {}
    for t in tiles:
        dt = timeit(lambda: ops.gemm(x, w, out=out, mode=ops.CONV3, conv=(n_img, h, w_, 1, 0), tile=t))
        r[t] = 2.0 * n_img * h * w_ * cout * 9 * cin / dt / 1e12
    return r

def bench_tmix(B, F, hw, C, tiles=(0, 4, 6, 9, 13, 17, 18, 20, 23, 24, 25, 26)):
    y = torch.randn(B * F * hw, C, device=dev).bfloat16(); w = (torch.randn(C, 3 * C, device=dev) * (3 * C) ** -0.5).bfloat16()
    out = torch.empty_like(y)
    r = {}
    for t in tiles:
        dt = timeit(lambda: ops.gemm(y, w, out=out, res1=y, mode=ops.TMIX, tmix=(hw, F), tile=t))
        r[t] = 2.0 * B * F * hw * C * 3 * C / dt / 1e12
    return r

fmt = lambda r: "  ".join(f"t{k}:{v:7.1f}" for k, v in r.items())
print("== plain (M,N,K) TFLOP/s per tile (0=auto,1=128x128,2=128x64,3=64x64)")
for M, N, K in [(24576, 320, 320), (24576, 320, 1280), (24576, 2560, 320), (6144, 640, 640), (6144, 5120, 640), (6144, 640, 2560),
                (1536, 1280, 1280), (1536, 10240, 1280), (1536, 1280, 5120), (384, 1280, 1280), (8192, 8192, 8192), (4096, 4096, 4096)]:
    print(f"plain {M:6d} {N:6d} {K:6d}  {fmt(bench_plain(M, N, K))}")
print(f"geglu 24576 2560 320     {fmt(bench_plain(24576, 2560, 320, geglu=True))}")
print("== conv3x3 (n_img,h,w,cin,cout)")
for n_img, h, w_, cin, cout in [(24, 32, 32, 320, 320), (24, 32, 32, 640, 320), (24, 32, 32, 960, 320), (24, 16, 16, 640, 640), (24, 16, 16, 1280, 640),
                                (24, 16, 16, 1920, 640), (24, 8, 8, 1280, 1280), (24, 8, 8, 2560, 1280), (24, 4, 4, 1280, 1280), (24, 4, 4, 2560, 1280)]:
    print(f"conv {n_img} {h}x{w_} {cin:5d}->{cout:5d}  {fmt(bench_conv(n_img, h, w_, cin, cout))}")
print("== tmix (B,F,hw,C)")
for B, F, hw, C in [(2, 12, 1024, 320), (2, 12, 256, 640), (2, 12, 64, 1280), (2, 12, 16, 1280)]:
    print(f"tmix {B} {F} {hw:5d} {C:5d}  {fmt(bench_tmix(B, F, hw, C))}")
