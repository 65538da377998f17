"""Every GEMM-family launch of one BASELINE cfg-2 UNet step against the vendor library on the same shape.

For each distinct (mode, M, N, K, epilogue) of the step: the launch as the step issues it (same descriptor and buffers, split-K
reduce included), replayed hot from a captured graph, next to torch.matmul (hipBLASLt) on a plain M x N x K problem of the same
size — for the temporal-mix and 3x3 implicit GEMMs that is the product WITHOUT the gather (a lower bound for a library path, which
would also have to materialise the gathered operand).  torch.matmul has no bias / residual / GEGLU / LayerNorm-fold epilogue
either.  The table is weighted by launches per step.  Yardstick only: the product never calls a BLAS library.
profiles/r3_step_vs_blas.txt is this script's output."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import asva_amd.unet as U  # noqa: E402
from asva_amd import ops  # noqa: E402
from asva_amd.conditioning import audio_segment_mask  # noqa: E402
from asva_amd.engine import DenoiseEngine  # noqa: E402
from asva_amd.schedulers import DDIMScheduler  # noqa: E402

dev = torch.device("cuda", 0)


def gtime(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def main():
    unet = bench.build_unet(dev, 0, 1)
    lat, text, audio, null_audio = bench.synthetic_clip(dev, 1000)
    eng = DenoiseEngine(unet, DDIMScheduler(), 4.0, use_graph=False)
    eng.set_conditioning(text, audio, null_audio, audio_segment_mask(12), 12)
    eng.prepare(lat, 50)
    for i in range(2):
        eng.step(lat, i)
    shapes = []
    orig_gemm = ops.gemm

    def gemm(a, w, **k):
        out = orig_gemm(a, w, **k)
        mode = k.get("mode", 0)
        K = {0: a.shape[1] + (k["a2"].shape[1] if k.get("a2") is not None else 0), 1: 3 * a.shape[1], 2: 9 * a.shape[1]}[mode]
        tag = ("plain", "tmix", "conv3")[mode] + ("+geglu" if k.get("geglu") else "") + ("+ln" if k.get("ln") is not None else "") + \
              ("+res" if k.get("res1") is not None else "")
        shapes.append((tag, out.shape[0], w.shape[0], K))
        return out

    orig_batched = ops.gemm_batched

    def gemm_batched(a, w, **k):       # the frame-0 K|V projection of the first-frame attention: B batches of L rows
        out = orig_batched(a, w, **k)
        shapes.append(("batched" + ("+ln" if k.get("ln") is not None else ""), a.shape[0] * a.shape[1], w.shape[1], a.shape[2]))
        return out

    class Proxy:
        def __getattr__(self, n):
            return gemm if n == "gemm" else gemm_batched if n == "gemm_batched" else getattr(ops, n)

    U.ops = Proxy()
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    unet.denoise_forward(lat, torch.full((1,), 501.0, device=dev), rep=2)
    ops.set_timer(None)
    torch.cuda.synchronize()
    U.ops = ops
    # ops.gemm and ops.gemm_batched register the gemm replays: one per wrapped call, in call order
    replays = [fn for fam, fn, _ in timer.replays if fam.startswith("gemm")]
    assert len(replays) == len(shapes), (len(replays), len(shapes))
    agg = collections.OrderedDict()
    for fn, key in zip(replays, shapes):
        d = agg.setdefault(key, {"n": 0, "fn": fn, "flops": 2.0 * key[1] * key[2] * key[3]})
        d["n"] += 1
    tot_own = tot_lib = tot_fl = 0.0
    rows = []
    for key, d in agg.items():
        tag, M, N, K = key
        t_own = gtime(d["fn"])
        if M:
            n_lib = N                      # (GEGLU: torch.matmul computes the full 2 x N/2 projection as well)
            a = torch.randn(M, K, device=dev).bfloat16()
            w = (0.02 * torch.randn(n_lib, K, device=dev)).bfloat16()
            out = torch.empty(M, n_lib, device=dev, dtype=torch.bfloat16)
            t_lib = gtime(lambda: torch.matmul(a, w.t(), out=out))
            del a, w, out
        else:
            t_lib = t_own
        rows.append((d["n"] * t_own, tag, M, N, K, d["n"], t_own, t_lib, d["flops"]))
        tot_own += d["n"] * t_own
        tot_lib += d["n"] * t_lib
        tot_fl += d["n"] * d["flops"]
    print(f"{'kind':18s} {'M':>6s} {'N':>6s} {'K':>6s} {'n':>3s} {'own us':>8s} {'lib us':>8s} {'own/lib':>8s} {'own TF/s':>9s}")
    for _, tag, M, N, K, n, t_own, t_lib, fl in sorted(rows, reverse=True):
        print(f"{tag:18s} {M:6d} {N:6d} {K:6d} {n:3d} {t_own:8.1f} {t_lib:8.1f} {t_own / t_lib:8.2f} {fl / t_own / 1e6:9.0f}")
    print(f"per step, hot, isolated: own kernels {tot_own / 1e3:.3f} ms ({tot_fl / tot_own / 1e6:.0f} TFLOP/s), "
          f"torch.matmul on the plain shapes {tot_lib / 1e3:.3f} ms ({tot_fl / tot_lib / 1e6:.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
