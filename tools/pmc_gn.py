"""The GroupNorm launches of a step (stats + apply pairs and the one-launch form, the shapes of tools/gn_bench.py) a few times each, for
rocprofv3 --pmc passes (profiles/r6_pmc_groupnorm.md):
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace ... -- python tools/pmc_gn.py
    rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum --kernel-trace ...
    rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace ...
summarised per (kernel, grid) by tools/pmc_table.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from asva_amd import _lib, ops  # noqa: E402

L = _lib.lib()
# (nb, rows per batch, c1, c2): the ResBlock norms (pooled over F, H, W; some on a skip concat) and the per-frame Transformer3D norms of a cfg-2 step
SHAPES = [(2, 12288, 320, 0), (2, 12288, 320, 320), (2, 12288, 640, 320), (24, 1024, 320, 0), (2, 3072, 640, 0), (2, 3072, 640, 640),
          (2, 3072, 1280, 640), (24, 256, 640, 0), (2, 768, 1280, 0), (2, 768, 1280, 1280), (24, 64, 1280, 0), (2, 192, 1280, 0), (2, 192, 1280, 1280)]
for nb, rows, c1, c2 in SHAPES:
    C = c1 + c2
    x1 = torch.randn(nb * rows, c1, device="cuda").bfloat16()
    x2 = torch.randn(nb * rows, c2, device="cuda").bfloat16() if c2 else None
    y = torch.empty(nb * rows, C, device="cuda", dtype=torch.bfloat16)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for fused in (False, True):
        if fused and not L.avsd_groupnorm_fused_supported(nb, rows, 32, c1, c2, 0):
            continue
        ops._GN_FUSED = fused
        for _ in range(5):
            ops.groupnorm(x1, x2, nb, rows, 32, g, b, 1e-5, True, out=y)
torch.cuda.synchronize()
