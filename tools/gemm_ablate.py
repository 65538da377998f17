import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from asva_amd import ops, _lib
orig = _lib.lib().avsd_gemm_bf16
extra = [0]
def hook(dref, stream):
    d = dref._obj
    d.flags |= extra[0]
    return orig(dref, stream)
class L:
    def __getattr__(self, n): return hook if n == "avsd_gemm_bf16" else getattr(_lib._lib, n)
real = _lib.lib()
_lib_lib = _lib.lib
_lib.lib = lambda: L()
for (M, N, K, tile, res) in [(24576, 320, 320, 13, True), (24576, 320, 320, 23, True), (1536, 1280, 1280, 7, False), (6144, 640, 640, 6, True), (24576, 320, 2880, 17, False)]:
    a = torch.randn(M, K if K != 2880 else 320, device="cuda").bfloat16(); w = (0.02 * torch.randn(N, K, device="cuda")).bfloat16()
    r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = f"{M}x{N}x{K} tile {tile}: "
    for name, fl in (("full", 0), ("no stores", 256), ("no k-loop", 512), ("neither", 768)):
        extra[0] = fl
        if K == 2880:
            f = lambda: ops.gemm(a, w, out=out, mode=ops.CONV3, conv=(24, 32, 32, 1, 0), tile=tile)
        else:
            f = lambda: ops.gemm(a, w, res1=r, out=out, tile=tile)
        for _ in range(3): f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        line += f"{name} {e0.elapsed_time(e1) / 100 * 1e3:6.1f} us | "
    print(line)
