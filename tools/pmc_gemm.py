"""Two representative GEMM launches for rocprofv3 --pmc runs: a memory-bound linear (24576x320x320) and the
32x32-level 3x3 conv (320->320), each launched 10x with fixed tiles (no autotune)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
from asva_amd.weights import pack_conv3x3
ops.set_autotune(False)
dev = "cuda"
M, C = 24576, 320
a = torch.randn(M, C, device=dev).bfloat16(); w = (torch.randn(C, C, device=dev) * C ** -0.5).bfloat16(); res = torch.randn(M, C, device=dev).bfloat16()
out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
wc = pack_conv3x3(torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for tile in (13, 4, 8):
    for _ in range(10):
        flush.zero_()                                   # evict L2 / infinity cache between launches
        ops.gemm(a, w, res1=res, out=out, tile=tile)
    for _ in range(10):
        flush.zero_()
        ops.gemm(a, wc, out=out, mode=ops.CONV3, conv=(24, 32, 32, 1, 0), tile=tile)
for tile in (13, 4, 8):
    for _ in range(10):                                 # warm: operands stay in the caches
        ops.gemm(a, w, res1=res, out=out, tile=tile)
torch.cuda.synchronize()
