"""The step's (and the VAE's low-resolution) 3x3 stride-1 convolutions: the tuned tap-major tile against every LDS-resident
tile (csrc/conv3r.hip) the geometry admits.  Hot, graph-timed (ops._time_hot), 16-bit output with bias + residual.
python tools/conv3r_bench.py [n_img,hs,ws,cin,cout ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_conv3x3

dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (24, 32, 32, 320, 320), (24, 32, 32, 640, 320), (24, 32, 32, 960, 320), (24, 32, 32, 640, 640),
    (24, 16, 16, 640, 640), (24, 16, 16, 1280, 640), (24, 16, 16, 1920, 640), (24, 16, 16, 960, 640), (24, 16, 16, 320, 640), (24, 16, 16, 1280, 1280),
    (24, 8, 8, 1280, 1280), (24, 8, 8, 2560, 1280), (24, 8, 8, 1920, 1280), (24, 8, 8, 640, 1280),
    (24, 4, 4, 1280, 1280), (24, 4, 4, 2560, 1280),
    (96, 32, 32, 320, 320), (96, 16, 16, 640, 640), (96, 8, 8, 1280, 1280), (96, 4, 4, 1280, 1280),     # 4 clips per forward
    (12, 32, 32, 512, 512),                                                                                   # VAE decoder, first level
    (12, 64, 64, 512, 512), (12, 128, 128, 512, 256), (12, 128, 128, 256, 256), (12, 256, 256, 256, 128), (12, 256, 256, 128, 128),   # VAE decoder, upper levels
    (48, 64, 64, 320, 320), (48, 64, 64, 640, 320), (48, 32, 32, 640, 640),                                 # cfg 4 (24 frames, 64 x 64 latents)
]
g = torch.Generator().manual_seed(0)
for n_img, hs, ws, cin, cout in shapes:
    M, K = n_img * hs * ws, 9 * cin
    x = torch.randn(M, cin, generator=g).to(torch.bfloat16).to(dev)
    w = pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev))
    b = torch.randn(cout, generator=g).to(dev)
    res = torch.randn(M, cout, generator=g).to(torch.bfloat16).to(dev)
    out = torch.empty(M, cout, dtype=torch.bfloat16, device=dev)
    flop = 2.0 * M * cout * K

    def run(t, sk):
        ops.gemm(x, w, bias=b, res1=res, out=out, mode=ops.CONV3, conv=(n_img, hs, ws, 1, 0), tile=t, split_k=sk)

    key = (ops.CONV3, M, cout, K)
    tuned = [v for k, v in ops.tile_cache().items() if tuple(k[:4]) == key]
    base = {}
    for cand in set(tuned) | {(20, 1), (30, 1), (32, 1) if cout == 320 and M % 96 == 0 else (9, 1)}:
        try:
            base[cand] = ops._time_hot(run, cand) * 1e3
        except Exception:  # noqa: BLE001
            pass
    bb = min(base, key=base.get)
    res_r = {}
    for cand in ops.conv3r_candidates(hs, ws, cin, M, cout):
        try:
            run(*cand)
            res_r[cand] = ops._time_hot(run, cand) * 1e3
        except Exception as e:  # noqa: BLE001
            res_r[cand] = float("inf")
            print("   ", cand, "failed:", str(e)[:80])
    line = f"{n_img:3d}x{hs:2d}x{ws:2d} cin {cin:4d} cout {cout:4d}  M {M:6d} K {K:6d} {flop / 1e9:6.1f} GF | tap-major {bb} {base[bb]:7.1f} us {flop / base[bb] / 1e6:6.0f} TF"
    if res_r:
        br = min(res_r, key=res_r.get)
        line += f" | resident {br} {res_r[br]:7.1f} us {flop / res_r[br] / 1e6:6.0f} TF  x{base[bb] / res_r[br]:.2f}"
        line += "   all: " + " ".join(f"{c[0]}/{c[1]}:{v:.0f}" for c, v in sorted(res_r.items(), key=lambda kv: kv[1])[:6])
    print(line, flush=True)
