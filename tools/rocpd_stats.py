"""Kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from a rocpd sqlite database."""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for n, s, e in rows:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"mom6hip_[a-z_0-9]+::\{lambda", "{lambda", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)( \[clone [^\]]*\])?$", "", n)[:120]
    a = agg.setdefault(n, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
out = csv.writer(open(sys.argv[2], "w")) if len(sys.argv) > 2 else None
hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]
if out:
    out.writerow(hdr)
print(f"{'kernel':80s} {'calls':>7s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if out:
        out.writerow([n, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]])
    print(f"{n[:80]:80s} {a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.2f} {100.0*a[1]/tot:6.2f}")
print(f"total kernel time {tot/1e6:.2f} ms")
