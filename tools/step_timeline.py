"""Steady-state view of a rocprofv3 kernel trace of bench.py (graph replay): takes the last `--steps` graph replays
(delimited by the guided_step kernel that ends each step) and reports, per kernel family, time per step, plus the idle
time between consecutive kernels."""
import sqlite3, sys
from collections import defaultdict
from prof_summary import short

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
nsteps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = sorted(cur.execute(f"select {name_col}, start, end from kernels").fetchall(), key=lambda r: r[1])
ends = [i for i, r in enumerate(rows) if "guided_step" in r[0]]
lo, hi = ends[-nsteps - 1] + 1, ends[-1] + 1
seg = rows[lo:hi]
def fam(n):
    k = short(n)
    for key in ("gemm4", "gemm2", "gemm1", "gemm_kernel", "conv3r", "nstream", "rowpanel", "splitk_reduce", "xattn", "attn_kernel", "tattn", "gn_", "layernorm", "linear_small", "guided", "ncfhw", "rows_to", "timestep"):
        if key in k: return {"gemm4": "gemm", "gemm2": "gemm", "gemm1": "gemm", "gemm_kernel": "gemm", "conv3r": "gemm", "nstream": "gemm", "rowpanel": "gemm"}.get(key, key)
    return k[:30]
agg = defaultdict(lambda: [0, 0.0]); busy = 0.0; gaps = 0.0; small = [0, 0.0]
for i, (n, s, e) in enumerate(seg):
    a = agg[fam(n)]; a[0] += 1; a[1] += (e - s) / 1e3; busy += (e - s) / 1e3
    if (e - s) < 8000: small[0] += 1; small[1] += (e - s) / 1e3
    if i: gaps += max(0, s - seg[i - 1][2]) / 1e3
wall = (seg[-1][2] - seg[0][1]) / 1e3
print(f"{nsteps} steps: wall {wall / nsteps / 1e3:.3f} ms/step, kernel-busy {busy / nsteps / 1e3:.3f} ms/step, idle between kernels {gaps / nsteps / 1e3:.3f} ms/step, {len(seg) / nsteps:.0f} launches/step")
print(f"kernels < 8 us: {small[0] / nsteps:.0f} per step, {small[1] / nsteps / 1e3:.3f} ms/step")
print("| family | launches/step | ms/step | avg us |\n|---|---:|---:|---:|")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| {k} | {n / nsteps:.0f} | {us / nsteps / 1e3:.3f} | {us / n:.1f} |")
