"""The step's temporal-mix GEMMs: tuned segment-major tile against the resident (12 frames x 32 pixels) tiles (csrc/conv3r.hip tmixr_kernel).
Hot, graph-timed.  python tools/tmixr_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for B, hw, C in ((2, 1024, 320), (2, 256, 640), (2, 64, 1280), (8, 1024, 320), (8, 256, 640), (8, 64, 1280), (2, 1024, 640), (2, 256, 1280)):
    Fr = 12
    M, K = B * Fr * hw, 3 * C
    y = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(C, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    out = torch.empty(M, C, dtype=torch.bfloat16, device=dev)

    def run(t, sk):
        ops.gemm(y, w, bias=b, res1=y, out=out, mode=ops.TMIX, tmix=(hw, Fr), tile=t, split_k=sk)

    tuned = [v for k, v in ops.tile_cache().items() if tuple(k[:4]) == (ops.TMIX, M, C, K)]
    base = {}
    for cand in set(tuned) | {(6, 1), (11, 1), (12, 1), (13, 1), (30, 1), (38, 1), (20, 1), (30, 2), (6, 4)}:
        try:
            base[cand] = ops._time_hot(run, cand) * 1e3
        except Exception:  # noqa: BLE001
            pass
    bb = min(base, key=base.get)
    res = {}
    for cand in ops.tmixr_candidates(hw, Fr, C, M, C):
        try:
            run(*cand)
            res[cand] = ops._time_hot(run, cand) * 1e3
        except Exception as e:  # noqa: BLE001
            print("   ", cand, "failed:", str(e)[:80])
    fl = 2.0 * M * C * K
    line = f"B {B} hw {hw:4d} C {C:4d}  M {M:6d} {fl / 1e9:6.1f} GF | segment-major {bb} {base[bb]:7.1f} us {fl / base[bb] / 1e6:5.0f} TF"
    if res:
        br = min(res, key=res.get)
        line += f" | resident {br} {res[br]:7.1f} us {fl / res[br] / 1e6:5.0f} TF  x{base[bb] / res[br]:.2f}   all: " + " ".join(f"{c[0]}/{c[1]}:{v:.0f}" for c, v in sorted(res.items(), key=lambda kv: kv[1])[:6])
    print(line, flush=True)
