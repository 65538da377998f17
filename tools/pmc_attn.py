"""The dominant attention launch (first-frame spatial attention at 32x32, d=40) a few times, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asva_amd import ops
B, F, C, L, heads = 2, 12, 320, 1024, 8
q = torch.randn(B * F * L, C, device="cuda").bfloat16()
kv = torch.randn(B * L, 2 * C, device="cuda").bfloat16()
out = torch.empty_like(q)
for _ in range(8):
    ops.attention(q, kv[:, :C], kv[:, C:], bq=B * F, lq=L, lk=L, kv_rows=L, heads=heads, q_per_kv=F, frames=F, out=out)
torch.cuda.synchronize()
