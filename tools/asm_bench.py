"""The hand-scheduled 4-wave tiles (csrc/gemm4.hip, ids 60-63) against the tuned LDS-direct tiles and torch.matmul (hipBLASLt, yardstick only):
bit-identity vs tile 9 on ragged shapes first, then hot graph-timed TFLOP/s.    python tools/asm_bench.py [--quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
ASM = [int(t) for t in os.environ.get("ASM_TILES", "60,61,62,63,64,65,66").split(",")]


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(torch.bfloat16).to(dev)


if "--x2" in sys.argv:
    # split precision: the asm tiles (63-66) against the LDS-direct X2 tile 11 (same three passes in the same order -> bit-identical)
    from asva_amd import precision as P
    P.set_split(True)
    bad = 0
    for M, N, K in [(256, 256, 64), (512, 384, 320), (1000, 640, 1280), (77, 132, 192), (3000, 320, 640)]:
        a, w = ops.to_act(torch.randn(M, K, generator=g).to(dev)), ops.to_act((torch.randn(N, K, generator=g) * K ** -0.5).to(dev))
        bias, res = torch.randn(N, generator=g).to(dev), ops.to_act(torch.randn(M, N, generator=g).to(dev))
        ref = ops.gemm(a, w, bias=bias, res1=res, tile=11)
        ref32 = ops.gemm(a, w, bias=bias, out_f32=True, tile=11)
        exact = ops.from_act(a).double() @ ops.from_act(w).double().T + bias.double()
        for t in (63, 64, 65, 66):
            out = ops.gemm(a, w, bias=bias, res1=res, tile=t)
            o32 = ops.gemm(a, w, bias=bias, out_f32=True, tile=t)
            same = torch.equal(ops.from_act(out), ops.from_act(ref)) and torch.equal(o32, ref32)
            e64 = ((o32.double() - exact).norm() / exact.norm()).item()
            bad += 0 if same else 1
            print(f"x2 parity {M}x{N}x{K} tile {t}: {'bit-identical to X2 tile 11' if same else 'DIFFERS'}  rel vs f64 {e64:.2e}", flush=True)
            if K >= 256:
                sk, rk = ops.gemm(a, w, bias=bias, out_f32=True, tile=t, split_k=2), ops.gemm(a, w, bias=bias, out_f32=True, tile=11, split_k=2)
                if not torch.equal(sk, rk):
                    bad += 1
                    print("   split_k=2 DIFFERS")
    for B, hw, C, N in [(2, 64, 320, 320), (1, 32, 128, 132), (1, 16, 1280, 1280)]:
        Fr = 12
        M = B * Fr * hw
        y, w = ops.to_act(torch.randn(M, C, generator=g).to(dev)), ops.to_act((torch.randn(N, 3 * C, generator=g) * (3 * C) ** -0.5).to(dev))
        b, res2 = torch.randn(N, generator=g).to(dev), ops.to_act(torch.randn(M, N, generator=g).to(dev))
        kw = dict(bias=b, res2=res2, mode=ops.TMIX, tmix=(hw, Fr))
        ref = ops.gemm(y, w, out_f32=True, tile=11, **kw)
        for t in (63, 64, 65, 66):
            for sk in (1, 2):
                o = ops.gemm(y, w, out_f32=True, tile=t, split_k=sk, **kw)
                r = ref if sk == 1 else ops.gemm(y, w, out_f32=True, tile=11, split_k=sk, **kw)
                if not torch.equal(o, r):
                    bad += 1
                    print(f"x2 tmix parity B={B} hw={hw} C={C} N={N} tile {t} split {sk}: DIFFERS rel {((o - r).norm() / r.norm()).item():.3e}", flush=True)
    print("X2 PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
    for M, N, K in [(24576, 320, 320), (24576, 320, 1280), (6144, 640, 640), (6144, 640, 2560), (1536, 1280, 1280), (1536, 1280, 5120), (24576, 2560, 320), (6144, 5120, 640), (384, 1280, 1280)]:
        a, w = ops.to_act(torch.randn(M, K, generator=g).to(dev)), ops.to_act((torch.randn(N, K, generator=g) * K ** -0.5).to(dev))
        b = torch.randn(N, generator=g).to(dev)
        out = ops.alloc16((M, N), dev)
        fl = 6.0 * M * N * K
        row = [f"x2 {M:6d}x{N:5d}x{K:5d}"]
        us = ops._time_hot(lambda tt, sk: ops.gemm(a, w, bias=b, out=out), (0, 1), reps=4) * 1e3
        row.append(f"table {us:7.1f} us {fl / us / 1e6:5.0f} TF(3p)")
        for t, sk in [(11, 1), (35, 1), (63, 1), (64, 1), (65, 1), (66, 1), (64, 2), (66, 2)]:
            if (K // 64) // sk < 4 and sk > 1:
                continue
            try:
                us = ops._time_hot(lambda tt, s_: ops.gemm(a, w, bias=b, out=out, tile=tt, split_k=s_), (t, sk), reps=4) * 1e3
                row.append(f"t{t}/{sk} {us:6.1f}")
            except Exception as e:  # noqa: BLE001
                row.append(f"t{t}/{sk} n/a")
        print("  ".join(row), flush=True)
    for B, hw, C, N in [(2, 64, 1280, 1280), (2, 16, 1280, 1280), (2, 256, 640, 640), (2, 1024, 320, 320)]:
        Fr = 12
        M = B * Fr * hw
        y, w = ops.to_act(torch.randn(M, C, generator=g).to(dev)), ops.to_act((torch.randn(N, 3 * C, generator=g) * (3 * C) ** -0.5).to(dev))
        b = torch.randn(N, generator=g).to(dev)
        out = ops.alloc16((M, N), dev)
        kw = dict(bias=b, res1=y, mode=ops.TMIX, tmix=(hw, Fr), out=out)
        row = [f"x2 tmix {M:6d}x{N:5d}x{3 * C:5d}"]
        us = ops._time_hot(lambda tt, sk: ops.gemm(y, w, **kw), (0, 1), reps=4) * 1e3
        row.append(f"table {us:7.1f} us")
        for t, sk in [(63, 1), (64, 1), (65, 1), (66, 1), (64, 2), (66, 2), (66, 4)]:
            try:
                us = ops._time_hot(lambda tt, s_: ops.gemm(y, w, tile=tt, split_k=s_, **kw), (t, sk), reps=4) * 1e3
                row.append(f"t{t}/{sk} {us:6.1f}")
            except Exception as e:  # noqa: BLE001
                row.append(f"t{t}/{sk} n/a")
        print("  ".join(row), flush=True)
    sys.exit(0)

# ---- parity: same products, same K order (rotated K walk off) -> bit-identical to the LDS-direct tile 9
bad = 0
ops.set_krot(False)
for M, N, K in [(256, 256, 64), (256, 256, 128), (512, 384, 320), (1000, 640, 1280), (77, 132, 192), (2048, 1280, 768), (3000, 320, 640)]:
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias, res = torch.randn(N, generator=g).to(dev), rnd(M, N)
    ref = ops.gemm(a, w, bias=bias, res1=res, tile=9)
    ref32 = ops.gemm(a, w, bias=bias, out_f32=True, tile=9)
    for t in ASM:
        out = ops.gemm(a, w, bias=bias, res1=res, tile=t)
        o32 = ops.gemm(a, w, bias=bias, out_f32=True, tile=t)
        same = torch.equal(out, ref) and torch.equal(o32, ref32)
        rep = all(torch.equal(ops.gemm(a, w, bias=bias, res1=res, tile=t), out) for _ in range(3))
        err = ((o32 - ref32).norm() / ref32.norm()).item()
        if not (same and rep):
            bad += 1
        print(f"parity {M}x{N}x{K} tile {t}: {'bit-identical' if same else f'DIFFERS rel {err:.3e}'}{'' if rep else '  NOT REPEATABLE'}", flush=True)
        if K >= 256:
            sk = ops.gemm(a, w, bias=bias, out_f32=True, tile=t, split_k=2)
            rk = ops.gemm(a, w, bias=bias, out_f32=True, tile=9, split_k=2)
            if not torch.equal(sk, rk):
                bad += 1
                print(f"   split_k=2 DIFFERS rel {((sk - rk).norm() / rk.norm()).item():.3e}")
# temporal-mix operands (segment jumps, split-K slices that start inside a segment), full epilogue
for B, hw, C, N in [(2, 64, 320, 320), (1, 32, 128, 132), (3, 96, 64, 64), (2, 32, 640, 640), (1, 16, 1280, 1280)]:
    Fr = 12
    M = B * Fr * hw
    y = rnd(M, C)
    w = rnd(N, 3 * C, scale=(3 * C) ** -0.5)
    b = torch.randn(N, generator=g).to(dev)
    temb = torch.randn(B, N, generator=g).to(dev)
    res2 = rnd(M, N)
    kw = dict(bias=b, rowvec=temb, rows_per_vec=Fr * hw, res2=res2, mode=ops.TMIX, tmix=(hw, Fr))
    ref = ops.gemm(y, w, out_f32=True, tile=9, **kw)
    for t in [t for t in ASM if t != 60]:
        for sk in (1, 2, 3):
            if (3 * C // 64) // sk < 2:
                continue
            o = ops.gemm(y, w, out_f32=True, tile=t, split_k=sk, **kw)
            r = ref if sk == 1 else ops.gemm(y, w, out_f32=True, tile=9, split_k=sk, **kw)
            ok = torch.equal(o, r)
            bad += 0 if ok else 1
            if not ok:
                print(f"tmix parity B={B} hw={hw} C={C} N={N} tile {t} split {sk}: DIFFERS rel {((o - r).norm() / r.norm()).item():.3e}", flush=True)
print("PARITY", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
ops.set_krot(True)
if "--tmix" in sys.argv:
    for B, hw, C, N in [(2, 64, 1280, 1280), (2, 16, 1280, 1280), (2, 256, 640, 640), (2, 1024, 320, 320), (2, 1024, 640, 640), (2, 256, 1280, 1280)]:
        Fr = 12
        M = B * Fr * hw
        y = rnd(M, C)
        w = rnd(N, 3 * C, scale=(3 * C) ** -0.5)
        b = torch.randn(N, generator=g).to(dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw = dict(bias=b, res1=y if N == C else None, mode=ops.TMIX, tmix=(hw, Fr), out=out)
        fl = 2.0 * M * N * 3 * C
        row = [f"tmix {M:6d}x{N:5d}x{3 * C:5d}"]
        us = ops._time_hot(lambda tt, sk: ops.gemm(y, w, **kw), (0, 1), reps=6) * 1e3
        row.append(f"table {us:7.1f} us")
        for t, sk in [(61, 1), (62, 1), (63, 1), (64, 1), (65, 1), (66, 1), (63, 2), (64, 2), (65, 2), (66, 2), (66, 4)]:
            if (3 * C // 64) // sk < 4:
                continue
            try:
                us = ops._time_hot(lambda tt, s_: ops.gemm(y, w, tile=tt, split_k=s_, **kw), (t, sk), reps=6) * 1e3
                row.append(f"asm{t}/{sk} {us:6.1f}")
            except Exception as e:  # noqa: BLE001
                row.append(f"asm{t}/{sk} n/a")
        print("  ".join(row), flush=True)
    sys.exit(0)

shapes = [(4096, 4096, 4096), (24576, 2560, 320), (6144, 5120, 640), (1536, 10240, 1280), (24576, 320, 1280), (6144, 640, 2560),
          (6144, 1920, 640), (24576, 960, 320), (1536, 1280, 5120), (6144, 640, 640), (1536, 1280, 1280), (24576, 320, 320), (384, 1280, 1280), (1536, 3840, 1280)]
if "--quick" in sys.argv:
    shapes = shapes[:5]
for M, N, K in shapes:
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    b = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    row = [f"{M:6d}x{N:5d}x{K:5d}"]
    torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    t_lib = ops._time_hot(lambda tt, sk: torch.matmul(a, w.t(), out=out), (0, 1), reps=6) * 1e3
    row.append(f"lib {t_lib:7.1f} us {fl / t_lib / 1e6:5.0f} TF")
    best = None
    for t in (7, 9, 11, 13, 14, 19, 20, 25, 30):
        try:
            us = ops._time_hot(lambda tt, sk: ops.gemm(a, w, bias=b, out=out, tile=tt), (t, 1), reps=6) * 1e3
            if best is None or us < best[0]:
                best = (us, t)
        except Exception:  # noqa: BLE001
            pass
    row.append(f"best v2 (tile {best[1]:2d}) {best[0]:7.1f} us {fl / best[0] / 1e6:5.0f} TF")
    for t in ASM:
        us = ops._time_hot(lambda tt, sk: ops.gemm(a, w, bias=b, out=out, tile=tt), (t, 1), reps=6) * 1e3
        row.append(f"asm{t} {us:7.1f} us {fl / us / 1e6:5.0f} TF")
    print("  ".join(row), flush=True)
