"""Whole-step PMC summary: three rocprofv3 --pmc databases of the same eager bench command
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d <d1> -o p -- python bench.py --no-graph --steps S --warmup 1 --no-cpu-baseline --no-roofline --no-vae
    rocprofv3 --pmc FETCH_SIZE  ... (same command)        rocprofv3 --pmc WRITE_SIZE ... (same command)
-> per kernel family and for the whole step: launches, MFMA-busy fraction, HBM-side bytes and GB/s (FETCH_SIZE doubled per
the gfx950 correction, MI355X_MICROARCH.md HBM section; FETCH/WRITE_SIZE are in KiB).  The denoising steps are delimited by
the guided_step kernel; the first `skip` steps (warm-up) are dropped.

    python tools/pmc_step.py <mfma.db> <fetch.db> <write.db> [--skip 1] > profiles/r2_pmc_step.md
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 256 CUs x 4 SIMDs); GRBM_GUI_ACTIVE is summed over the 8
XCDs by rocprofv3, hence the / 8.  Durations come from the kernel trace of the MFMA pass (a profiled pass clocks ~3-5 %
lower than an un-profiled run: compare fractions, not absolute times)."""
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short  # noqa: E402


def fam(n):
    k = short(n)
    for key, name in (("gemm4", "gemm"), ("gemm2", "gemm"), ("gemm1", "gemm"), ("conv3r", "gemm"), ("nstream", "gemm"), ("splitk_reduce", "splitk_reduce"), ("xattn", "fused_cross_attention"),
                      ("mlp_", "fused_mlp"), ("attn_kernel", "attention"), ("attn_f8", "attention"), ("tattn", "temporal_attention"), ("gn_", "groupnorm"),
                      ("layernorm", "layernorm"), ("linear_small", "small"), ("guided", "small"), ("ncfhw", "small"),
                      ("rows_to", "small"), ("timestep", "small"), ("copy_rep", "small")):
        if key in k:
            return name
    return "other"


def load(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ci = {c: i for i, c in enumerate(cols)}
    per = defaultdict(lambda: defaultdict(float))
    names, order = {}, {}
    for r in cur.execute("select * from counters_collection"):
        d = r[ci["dispatch_id"]]
        per[d][r[ci["counter_name"]]] += r[ci["value"]]
        names[d] = r[ci["kernel_name"]]
    kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    dur = {}
    if kcols:
        name_col = "name" if "name" in kcols else [c for c in kcols if "name" in c][0]
        did = "dispatch_id" if "dispatch_id" in kcols else None
        if did:
            for d, s, e in cur.execute(f"select {did}, start, end from kernels"):
                dur[d] = (e - s) / 1e3
    return per, names, dur


def steps_of(names, skip):
    ids = sorted(names)
    ends = [i for i, d in enumerate(ids) if "guided_step" in names[d]]
    lo = ends[skip - 1] + 1 if skip > 0 else 0
    return ids[lo:ends[-1] + 1], len(ends) - skip


def main():
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 1
    mf, names_m, dur = load(sys.argv[1])
    ft, names_f, _ = load(sys.argv[2])
    wr, names_w, _ = load(sys.argv[3])
    ids_m, n = steps_of(names_m, skip)
    ids_f, n_f = steps_of(names_f, skip)
    ids_w, n_w = steps_of(names_w, skip)
    assert n == n_f == n_w and len(ids_m) == len(ids_f) == len(ids_w), (n, n_f, n_w, len(ids_m), len(ids_f), len(ids_w))
    agg = defaultdict(lambda: defaultdict(float))
    for dm, df, dw in zip(ids_m, ids_f, ids_w):
        f = fam(names_m[dm])
        assert fam(names_f[df]) == f == fam(names_w[dw])
        a = agg[f]
        a["n"] += 1
        a["mfma"] += mf[dm].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a["gui"] += mf[dm].get("GRBM_GUI_ACTIVE", 0.0)
        a["fetch"] += 2.0 * 1024.0 * ft[df].get("FETCH_SIZE", 0.0)
        a["write"] += 1024.0 * wr[dw].get("WRITE_SIZE", 0.0)
        a["us"] += dur.get(dm, 0.0)
    tot = defaultdict(float)
    print(f"{n} eager steps, {sum(a['n'] for a in agg.values()) / n:.0f} launches per step\n")
    print("| family | launches/step | ms/step (profiled) | MFMA busy | HBM-side MB/step (fetch x2 + write) | GB/s |")
    print("|---|---:|---:|---:|---:|---:|")
    for f, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        busy = a["mfma"] / (a["gui"] / 8.0 * 256 * 4) if a["gui"] else 0.0
        mb = (a["fetch"] + a["write"]) / n / 1e6
        gbs = (a["fetch"] + a["write"]) / (a["us"] * 1e-6) / 1e9 if a["us"] else 0.0
        print(f"| {f} | {a['n'] / n:.0f} | {a['us'] / n / 1e3:.3f} | {100 * busy:.1f} % | {mb:.0f} | {gbs:.0f} |")
        for k in ("n", "mfma", "gui", "fetch", "write", "us"):
            tot[k] += a[k]
    busy = tot["mfma"] / (tot["gui"] / 8.0 * 256 * 4)
    print(f"| **whole step** | {tot['n'] / n:.0f} | {tot['us'] / n / 1e3:.3f} | {100 * busy:.1f} % | {(tot['fetch'] + tot['write']) / n / 1e6:.0f} | "
          f"{(tot['fetch'] + tot['write']) / (tot['us'] * 1e-6) / 1e9:.0f} |")
    print(f"\nMFMA instructions per step: {tot['mfma'] / 32 / n:.3e} (v_mfma_f32_32x32x16 = 32768 FLOP each -> "
          f"{tot['mfma'] / 32 / n * 32768 / 1e12:.3f} TFLOP executed per step)")


if __name__ == "__main__":
    main()
