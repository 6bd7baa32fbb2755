"""Per-kernel averages of the PMC counters in a rocprofv3 (rocpd sqlite) database."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
views = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
if len(sys.argv) > 2 and sys.argv[2] == "schema":
    for v in views:
        if "pmc" in v or "counter" in v:
            print(v, [r[1] for r in c.execute(f"pragma table_info({v})")])
    sys.exit(0)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select kernel_name, counter_name, value from counters_collection").fetchall() if "kernel_name" in cols else []
agg = {}
for n, cn, v in rows:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)( \[clone [^\]]*\])?$", "", n)[:70]
    a = agg.setdefault((n, cn), [0, 0.0]); a[0] += 1; a[1] += v
# the 40 largest kernels of every counter (a listing cut over all counters together loses the instruction counts behind the cycle counts)
for counter in sorted({cn for _, cn in agg}):
    mine = [(k, a) for k, a in agg.items() if k[1] == counter]
    for (n, cn), a in sorted(mine, key=lambda kv: -kv[1][1])[:40]:
        print(f"{n:70s} {cn:12s} dispatches {a[0]:5d}  avg {a[1]/a[0]:16.1f}  total {a[1]:18.1f}")
