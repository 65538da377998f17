"""A few back-to-back launches of chosen GEMM shapes with fixed tiles, for rocprofv3 --pmc passes (cache latency / stall counters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
dev = "cuda"
for M, N, K, tile, sk in [(1536, 1280, 1280, 4, 1), (24576, 320, 320, 13, 1), (6144, 640, 640, 6, 1), (4096, 4096, 4096, 9, 1)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (0.05 * torch.randn(N, K, device=dev)).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(6):
        ops.gemm(a, w, out=out, tile=tile, split_k=sk)
    torch.cuda.synchronize()
