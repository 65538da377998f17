"""Which hipBLASLt kernel does torch.matmul pick on the step's GEMM shapes?  Yardstick intel only (the product never calls it).
    cd /tmp && rocprofv3 --kernel-trace -d /tmp/libk -o k -- python tools/lib_kernels.py run
    python tools/lib_kernels.py show /tmp/libk/.../k_results.db
`run` issues every shape REPS times in a fixed order; `show` walks the trace in start order and prints one line per shape."""
import sys

SHAPES = [(24576, 2560, 320), (6144, 5120, 640), (1536, 10240, 1280), (1536, 1280, 3840), (384, 1280, 3840), (1536, 1280, 1280),
          (6144, 640, 640), (6144, 640, 1920), (24576, 320, 960), (6144, 640, 5760), (24576, 320, 320), (24576, 320, 1280),
          (1536, 1280, 5120), (6144, 640, 2560), (6144, 1920, 640), (384, 1280, 1280), (8192, 8192, 8192)]
REPS = 6


def run():
    import torch
    dev = torch.device("cuda", 0)
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (0.02 * torch.randn(N, K, device=dev)).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(REPS):
            torch.matmul(a, w.t(), out=out)
        torch.cuda.synchronize()


def show(path):
    import sqlite3
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    extra = [c for c in ("grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "lds_size", "lds_block_size", "vgpr_count", "arch_vgpr_count", "accum_vgpr_count", "sgpr_count") if c in cols]
    rows = cur.execute(f"select {name_col}, start, end{''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
    gem = [r for r in rows if "Cijk" in r[0] or "gemm" in r[0].lower()]
    print("columns:", cols)
    i = 0
    for M, N, K in SHAPES:
        grp = gem[i:i + REPS]
        i += REPS
        if not grp:
            break
        us = sorted((r[2] - r[1]) / 1e3 for r in grp)
        print(f"{M:6d} {N:6d} {K:6d}  med {us[len(us) // 2]:7.1f} us  {dict(zip(extra, grp[-1][3:]))}\n    {grp[-1][0]}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        show(sys.argv[2])
