"""The 256 x 256 phase-interleaved GEMM tile (csrc/gemm8p.hip, tile id 37): correctness against an f32 matmul / conv on
ragged and aligned shapes, a run-to-run race screen, and GPU time against the library's other big tiles and torch.matmul
(hipBLASLt).  profiles/r3_8phase_probe.txt is this script's output."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from asva_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
T8 = ops.TILE_8PHASE


def gtime(fn, reps=10):
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
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def check():
    ok = True
    gen = torch.Generator(device="cpu").manual_seed(0)
    for M, N, K in [(256, 256, 64), (256, 256, 128), (256, 256, 448), (512, 512, 1024), (300, 260, 200), (1000, 520, 328), (2048, 1280, 640)]:
        a = torch.randn(M, K, generator=gen).to(dev).bfloat16()
        w = (torch.randn(N, K, generator=gen) * K ** -0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=gen).to(dev)
        res = torch.randn(M, N, generator=gen).to(dev).bfloat16()
        ref = a.float() @ w.float().T + bias + res.float()
        out = ops.gemm(a, w, bias=bias, res1=res, tile=T8)
        base = ops.gemm(a, w, bias=bias, res1=res, tile=9)
        e = rel(out, ref)
        same = bool(torch.equal(out, base))
        runs = [ops.gemm(a, w, bias=bias, res1=res, tile=T8) for _ in range(20)]
        stable = all(torch.equal(r, out) for r in runs)
        print(f"plain {M}x{N}x{K}: rel {e:.2e}  == tile 9: {same}  20 runs identical: {stable}")
        ok &= e < 5e-3 and stable
    for n_img, hs, cin, cout, stride, ups in [(2, 32, 64, 256, 1, 0), (3, 24, 128, 320, 1, 0), (2, 32, 192, 256, 2, 0), (2, 16, 128, 256, 1, 1)]:
        x = torch.randn(n_img * hs * hs, cin, generator=gen).to(dev).bfloat16()
        wt = (torch.randn(cout, 3, 3, cin, generator=gen) * (9 * cin) ** -0.5).to(dev).bfloat16()
        xi = x.float().reshape(n_img, hs, hs, cin).permute(0, 3, 1, 2)
        if ups:
            xi = F.interpolate(xi, scale_factor=2, mode="nearest")
        ref = F.conv2d(xi, wt.float().permute(0, 3, 1, 2), stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
        out = ops.gemm(x, wt.reshape(cout, -1), mode=ops.CONV3, conv=(n_img, hs, hs, stride, ups), tile=T8)
        base = ops.gemm(x, wt.reshape(cout, -1), mode=ops.CONV3, conv=(n_img, hs, hs, stride, ups), tile=9)
        e = rel(out, ref)
        print(f"conv3 {n_img}x{hs}x{hs} cin {cin} -> {cout} stride {stride} ups {ups}: rel {e:.2e}  == tile 9: {bool(torch.equal(out, base))}")
        ok &= e < 5e-3
    return ok


def speed():
    for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192), (6144, 5120, 640), (24576, 2560, 320), (98304, 320, 2880), (24576, 1280, 640), (196608, 256, 1152), (49152, 512, 4608)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (0.02 * torch.randn(N, K, device=dev)).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        t_lib = gtime(lambda: torch.matmul(a, w.t(), out=out))
        line = f"{M:6d} {N:5d} {K:5d}: torch.matmul {t_lib:7.1f} us ({fl / t_lib / 1e6:5.0f} TF)"
        for tile in (9, 20, 23, 31, T8):
            try:
                t = gtime(lambda: ops.gemm(a, w, out=out, tile=tile))
                line += f"  tile {tile:2d} {t:7.1f} ({fl / t / 1e6:5.0f})"
            except Exception as e:
                line += f"  tile {tile}: {str(e)[:30]}"
        print(line, flush=True)


if __name__ == "__main__":
    good = check()
    print("correct:", good)
    if good or os.environ.get("P8_FORCE"):
        speed()
