"""The attention-family launches of a step (first-frame attention at 32x32 / 16x16 / 8x8, temporal attention at the four levels) a few
times each, for rocprofv3 --pmc passes; the fused cross-attention block comes from tools/xattn_bench.py --reps 3 --no-graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from asva_amd import ops

B, F, heads = 2, 12, 8
for C, L in ((320, 1024), (640, 256), (1280, 64)):
    q = torch.randn(B * F * L, C, device="cuda").bfloat16()
    kv = torch.randn(B * L, 2 * C, device="cuda").bfloat16()
    out = torch.empty_like(q)
    for _ in range(5):
        ops.attention(q, kv[:, :C], kv[:, C:], bq=B * F, lq=L, lk=L, kv_rows=L, heads=heads, q_per_kv=F, frames=F, out=out)
for C, hw in ((320, 1024), (640, 256), (1280, 64), (1280, 16)):
    qkv = torch.randn(B * F * hw, 3 * C, device="cuda").bfloat16()
    out = torch.empty(B * F * hw, C, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.temporal_attention(qkv, b=B, frames=F, hw=hw, heads=heads, out=out)
torch.cuda.synchronize()
