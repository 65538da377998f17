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
    extra = ""
    if os.environ.get("YARDSTICK_PRECISE", "1") != "0" and M * N <= 8192 * 8192:
        # the exact-f32 MFMA yardstick and the split-precision (3-pass) GEMM on the same shape
        af, wf = a.float(), w.float()
        t_f32 = gtime(lambda: ops.gemm_f32(af, wf), reps=5)
        from asva_amd import precision as P
        P.set_split(True)
        ops.set_autotune(False)
        a2, w2 = ops.to_act(af), ops.to_act(wf)
        o2 = ops.alloc16((M, N), a2.device)
        t_x2 = gtime(lambda: ops.gemm(a2, w2, out=o2), reps=5)
        P.set_split(False)
        extra = f"  | exact-f32 MFMA {t_f32:8.1f} us ({fl / t_f32 / 1e6:5.0f} TF)  split x2 {t_x2:7.1f} us ({3 * fl / t_x2 / 1e6:5.0f} TF of 3-pass work)"
    print(f"{M:6d} {N:6d} {K:6d}: torch.matmul {t_lib:7.1f} us ({fl / t_lib / 1e6:6.0f} TF)   avsd_gemm rule {t_rule:7.1f} us ({fl / t_rule / 1e6:6.0f} TF)  tuned {t_own:7.1f} us ({fl / t_own / 1e6:6.0f} TF){extra}")
