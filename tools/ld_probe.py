"""Does the leading dimension matter?  A GEMM tile stage reads 64-256 row segments of 128 B whose addresses are one row
stride apart; the UNet's strides are multiples of 512 B (K = 320 ... 11520 16-bit elements).  If the L2 / memory channels
are selected by low address bits, such strides fold all segments of a stage onto a few channels.  This probe times the
low-resolution GEMM shapes with W (and A) padded by 0 / 64 / 128 / 192 elements per row, hot (graph replay on one buffer)
and cold (rotating over enough copies of W to exceed L2 + Infinity Cache).  profiles/r3_ld_probe.txt is its output."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from asva_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn_list, reps=3):
    """fn_list: launches executed round-robin inside one graph; -> us per launch"""
    for f in fn_list:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fn_list:
                f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * len(fn_list)) * 1e3)
    return best


def main():
    shapes = [(384, 1280, 1280), (384, 1280, 3840), (384, 1280, 11520), (1536, 1280, 1280), (1536, 1280, 3840), (1536, 1280, 11520),
              (1536, 10240, 1280), (6144, 640, 1920), (6144, 640, 5760), (24576, 320, 960), (24576, 320, 2880)]
    for M, N, K in shapes:
        line = f"{M:6d} x {N:5d} x {K:5d}:"
        base = {}
        for pad_w, pad_a in ((0, 0), (64, 0), (128, 0), (192, 0), (0, 64), (64, 64), (192, 192)):
            ncopy = max(2, int(600e6 // (N * (K + pad_w) * 2)) + 1)
            ws = [(torch.randn(N, K + pad_w, device=dev) * 0.05).bfloat16()[:, :K] for _ in range(min(ncopy, 48))]
            a = torch.randn(M, K + pad_a, device=dev).bfloat16()[:, :K]
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            kw = {}          # tile from the shipped table / rule: the key does not depend on the strides
            hot = timed([lambda: ops.gemm(a, ws[0], out=out, **kw)], reps=20)
            cold = timed([(lambda w=w: ops.gemm(a, w, out=out, **kw)) for w in ws], reps=2)
            if (pad_w, pad_a) == (0, 0):
                base = dict(hot=hot, cold=cold)
            line += f"  W+{pad_w:<3d} A+{pad_a:<3d} hot {hot:6.1f} cold {cold:6.1f}"
            del ws
        print(line, flush=True)


if __name__ == "__main__":
    main()
