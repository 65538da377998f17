"""Two rocprofv3 --pmc databases (FETCH_SIZE pass, WRITE_SIZE pass) of `bench.py --no-graph` -> profiles/pmc_traffic.json:
mean HBM-side bytes per GEMM-family launch.  FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled (the
gfx950 rocprofv3 tallies 128-B read requests at 64 B — MI355X_MICROARCH.md, HBM section)."""
import json, sqlite3, sys
from collections import defaultdict

def per_dispatch(db_path, counter):
    db = sqlite3.connect(db_path); cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ci = {c: i for i, c in enumerate(cols)}
    acc = defaultdict(float); names = {}
    for r in cur.execute("select * from counters_collection"):
        if r[ci["counter_name"]] != counter: continue
        acc[r[ci["dispatch_id"]]] += r[ci["value"]]
        names[r[ci["dispatch_id"]]] = r[ci["kernel_name"]]
    return acc, names

fetch, names_f = per_dispatch(sys.argv[1], "FETCH_SIZE")
write, names_w = per_dispatch(sys.argv[2], "WRITE_SIZE")
isg = lambda n: ("gemm_kernel" in n or "gemm2_kernel" in n or "gemm4_kernel" in n or "conv3r_kernel" in n or "conv3r2d_kernel" in n or "nstream_kernel" in n)
gf = [v for d, v in fetch.items() if isg(names_f[d])]
gw = [v for d, v in write.items() if isg(names_w[d])]
assert gf and len(gf) == len(gw), (len(gf), len(gw))
fb = 2.0 * 1024.0 * sum(gf) / len(gf)
wb = 1024.0 * sum(gw) / len(gw)
out = {"hbm_bytes_per_launch": round(fb + wb), "fetch_bytes_x2_per_launch": round(fb), "write_bytes_per_launch": round(wb),
       "launches_profiled": len(gf),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python bench.py --no-graph "
                 "--steps 2 --warmup 1 --no-cpu-baseline --no-roofline, tuned tiles preloaded; tools/pmc_traffic.py"}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["build_id"] = bench.build_id()      # ties the counters to the kernel library + tile table they were collected from
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
