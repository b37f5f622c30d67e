"""Turn a rocprofv3 results database (rocpd sqlite, `--kernel-trace --stats`) into the per-kernel
summary CSV committed under profiles/.  usage: rocpd_summary.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
print(f"{len(rows)} kernels -> {out}")
