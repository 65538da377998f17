"""rocprofv3 --pmc rocpd database -> one line per (kernel, grid size): mean of every counter over the dispatches."""
import sqlite3, sys
from collections import defaultdict
from prof_summary import short
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
ci = {c: i for i, c in enumerate(cols)}
agg = defaultdict(lambda: defaultdict(float))
for r in cur.execute("select * from counters_collection"):
    agg[(short(r[ci["kernel_name"]]), r[ci["grid_size"]], r[ci["dispatch_id"]])][r[ci["counter_name"]]] += r[ci["value"]]
per = defaultdict(lambda: defaultdict(list))
for (k, g, d), cs in agg.items():
    for c, v in cs.items():
        per[(k, g)][c].append(v)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (k, g), cs in per.items():
    if flt in k:
        print(k, "grid", g, {c: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for c, v in cs.items()})
