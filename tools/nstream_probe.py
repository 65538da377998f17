"""The GEGLU projection (LayerNorm fold + bias + value * gelu(gate)) on the A-resident N-streaming tile (70, csrc/nstream.hip) against
the table's tile and the library's plain product of all 8 C columns: hot, graph-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_frag, pack_geglu

STAMPS = "--stamps" in sys.argv        # AVSD_LIB_PATH = a library built with -DNS_STAMPS: per-wave timeline of 4 workgroups
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for M, C in [(24576, 320), (6144, 640), (12288, 320), (98304, 320), (24576, 640)]:
    x = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(8 * C, C, generator=g) * C ** -0.5).to(dev)
    b = torch.randn(8 * C, generator=g).to(dev)
    wp, bp = pack_geglu(w, b)
    wf = pack_frag(wp)
    stats = torch.empty(M, C // 32, 2, device=dev)
    h = ops.gemm(x, torch.eye(C, device=dev).to(torch.bfloat16), rowstats=stats)
    cs = wp.float().sum(1)
    out = torch.empty(M, 4 * C, dtype=torch.bfloat16, device=dev)
    folded = ops.ln_fold(stats)
    t_fold = ops._time_hot(lambda *_: ops.ln_fold(stats), ()) * 1e3
    ops._NSTREAM = False
    ref = ops.gemm(h, wp, bias=bp, geglu=True, ln=(folded, cs, 1e-5), out=out).clone()
    t_tab = ops._time_hot(lambda *_: ops.gemm(h, wp, bias=bp, geglu=True, ln=(folded, cs, 1e-5), out=out), ()) * 1e3
    ops._NSTREAM = True
    got = ops.gemm(h, wp, bias=bp, geglu=True, ln=(stats, cs, 1e-5), out=out, w_frag=wf)
    same = torch.equal(got, ref)
    t_ns = ops._time_hot(lambda *_: ops.gemm(h, wp, bias=bp, geglu=True, ln=(stats, cs, 1e-5), out=out, w_frag=wf), ()) * 1e3
    if STAMPS and M == 24576 and C == 320:
        import ctypes, numpy as np
        from asva_amd import _lib
        torch.cuda.synchronize()
        for _ in range(3):
            ops.gemm(h, wp, bias=bp, geglu=True, ln=(stats, cs, 1e-5), out=out, w_frag=wf)
        torch.cuda.synchronize()
        buf = np.zeros(4 * 8 * 64, dtype=np.uint64)
        lib = ctypes.CDLL(os.environ["AVSD_LIB_PATH"])
        lib.avsd_nstream_debug_read(ctypes.c_void_p(buf.ctypes.data))
        d = buf.reshape(4, 8, 64).astype(np.int64)
        t0 = d[:, :, 0].min()
        for wg in range(4):
            print(f"workgroup {wg}: start {d[wg, :, 0].min() - t0} cycles after the first")
            for wv in range(8):
                r = d[wg, wv]
                T = 10
                loops = [int(r[3 + 3 * t] - r[2 + 3 * t]) for t in range(T)]
                epis = [int(r[4 + 3 * t] - r[3 + 3 * t]) for t in range(T)]
                print(f"  wave {wv}: prologue {int(r[1] - r[0]):6d}  K loops {loops}  epilogues {epis}  total {int(r[4 + 3 * (T - 1)] - r[0])}")
    a_l, w_l = torch.randn(M, C, device=dev).bfloat16(), torch.randn(8 * C, C, device=dev).bfloat16()
    o_l = torch.empty(M, 8 * C, dtype=torch.bfloat16, device=dev)
    torch.matmul(a_l, w_l.t(), out=o_l)
    torch.cuda.synchronize()
    t_lib = ops._time_hot(lambda *_: torch.matmul(a_l, w_l.t(), out=o_l), ()) * 1e3
    fl = 2.0 * M * 8 * C * C
    print(f"M={M:6d} C={C:4d}: table tile {t_tab:6.1f} us (+ ln_fold {t_fold:4.1f})   nstream {t_ns:6.1f} us ({fl / t_ns / 1e6:6.0f} TFLOP/s)   "
          f"torch.matmul plain {t_lib:6.1f} us   bit-identical to the table's tile: {same}", flush=True)
