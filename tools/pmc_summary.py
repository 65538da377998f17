"""Summarise rocprofv3 --pmc output (rocpd sqlite): mean counter value per kernel name (+ duration when traced)."""
import re, sqlite3, sys
from collections import defaultdict
from prof_summary import short
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print("cols:", cols)
rows = cur.execute("select * from counters_collection").fetchall()
ci = {c: i for i, c in enumerate(cols)}
agg = defaultdict(lambda: defaultdict(list))
order = []
for r in rows:
    k = short(r[ci["kernel_name"]]) if "kernel_name" in ci else str(r[ci.get("name", 0)])
    agg[(k, r[ci["dispatch_id"]])][r[ci["counter_name"]]].append(r[ci["value"]])
per = defaultdict(lambda: defaultdict(list))
for (k, d), cs in agg.items():
    for c, v in cs.items():
        per[k][c].append(sum(v))
for k, cs in per.items():
    if (sys.argv[2] if len(sys.argv) > 2 else "gemm") not in k: continue
    print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in cs.items()})
