"""Yardstick only (not used by the product): the same plain GEMM shapes through torch.matmul (hipBLASLt / rocBLAS) and
through avsd_gemm_bf16, GPU time from captured graphs.  Two columns for the hand-written kernel: the tile the static rule of
ops._heuristic_tile picks for a shape that is not in the committed table (what an unseen geometry gets), and the tile the
measuring tuner picks (what the table holds for the UNet's own shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
def gtime(fn, reps=20):
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
for M, N, K in [(24576, 320, 320), (24576, 320, 1280), (24576, 2560, 320), (6144, 640, 640), (6144, 5120, 640), (1536, 1280, 1280), (1536, 10240, 1280),
                (1536, 1280, 5120), (384, 1280, 1280), (384, 1280, 11520), (1536, 1280, 11520), (4096, 4096, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (0.02 * torch.randn(N, K, device="cuda")).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_lib = gtime(lambda: torch.matmul(a, w.t(), out=out))
    ops.set_autotune(False)
    t_rule = gtime(lambda: ops.gemm(a, w, out=out))
    ops.set_autotune(True)
    ops.gemm(a, w, out=out)                 # tunes this shape
    t_own = gtime(lambda: ops.gemm(a, w, out=out))
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:6d} {K:6d}: torch.matmul {t_lib:7.1f} us ({fl / t_lib / 1e6:6.0f} TF)   avsd_gemm rule {t_rule:7.1f} us ({fl / t_rule / 1e6:6.0f} TF)  tuned {t_own:7.1f} us ({fl / t_own / 1e6:6.0f} TF)")
