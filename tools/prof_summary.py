"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count, total/avg time, share.
    python tools/prof_summary.py gpurun_out/prof_xxx/name_results.db [--steps N] > profiles/xxx.md
Template arguments of the gemm kernels are decoded into readable tile names."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    modes = ("plain", "tmix", "conv3", "conv3-subpixel")
    m = re.search(r"gemm2_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, (\w+))?(?:, (\d+))?>", name)
    if m:
        bm, bn, wm, wn, st, mode, lw, x2, ex = m.groups()
        return (f"gemm2<{bm}x{bn},{wm}x{wn}w{'+' + lw + 'L' if lw and lw != '0' else ''},{st}st,{modes[int(mode)]}"
                f"{',x2' if x2 in ('true', '1') else ''}{',+rest' if ex == '1' else ''}>")
    m = re.search(r"conv3r_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\w+))?>", name)
    if m:
        bm, bn, wm, wn, st, lw, gn = m.groups()
        return f"conv3r<{bm}x{bn},{wm}x{wn}w{'+' + lw + 'L' if lw != '0' else ''},{st}st{',gn' if gn in ('true', '1') else ''}>"
    m = re.search(r"gemm4_kernel<(\d+), (\d+), (\d+)(?:, (\w+))?(?:, (\d+))?(?:, (\d+))?>", name)
    if m:
        fm, fn, mode, x2, _ns, ex = m.groups()
        return f"gemm4<{64 * int(fm)}x{64 * int(fn)},asm,{modes[int(mode)]}{',x2' if x2 in ('true', '1') else ''}{',+rest' if ex == '1' else ''}>"
    m = re.search(r"nstream_kernel<(\d+), (\d+), (\w+)>", name)
    if m:
        k, rf, fast = m.groups()
        return f"nstream<A-resident 96 rows x K={k}, W streamed{',geglu' if fast in ('true', '1') else ''}>"
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d+)(?:, (\d+))?>", name)
    if m:
        bm, bn, mode, ex = m.groups()
        return f"gemm1<{bm}x{bn},{modes[int(mode)]}{',+rest' if ex == '1' else ''}>"
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    by_grid = "--by-grid" in sys.argv          # one row per (kernel, grid): the grid tells the shapes of one kernel apart
    gcols = [c for c in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if c in cols] if by_grid else []
    if by_grid and not gcols:
        print("columns of `kernels`:", cols, file=sys.stderr)
    rows = cur.execute(f"select {name_col}, start, end" + "".join(", " + c for c in gcols) + " from kernels").fetchall()
    agg = {}
    for name, s, e, *g in rows:
        k = short(name) + (" grid " + "x".join(str(v) for v in g) if g else "")
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    total = sum(v[1] for v in agg.values())
    print(f"| kernel | launches | total us | avg us | share |")
    print(f"|---|---:|---:|---:|---:|")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:(120 if by_grid else 40)]:
        print(f"| `{k}` | {n} | {us:.0f} | {us / n:.2f} | {100 * us / total:.1f}% |")
    print(f"\ntotal kernel time {total / 1e3:.2f} ms over {sum(v[0] for v in agg.values())} launches" +
          (f"; {total / 1e3 / steps:.3f} ms per step over {steps} steps (all kernels incl. warm-up/autotune)" if steps else ""))


if __name__ == "__main__":
    main()
