"""Summarise a rocprofv3 --pmc results .db: per kernel, average counter values per dispatch."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
try:
    rows = cur.execute("select * from counters_collection limit 1").fetchall()
    cols = [d[0] for d in cur.description]
    print("counters_collection cols:", cols)
    q = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
except Exception as e:
    print("fallback", e)
    q = []
agg = collections.defaultdict(dict)
for k, c, v, n in q:
    agg[k][c] = (v, n)
for k, d in agg.items():
    if "icon" not in k: continue
    print(k[:90])
    for c, (v, n) in sorted(d.items()):
        print(f"    {c:28s} {v:18.1f}  (n={n})")
