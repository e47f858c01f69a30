#!/usr/bin/env python3
"""Export the per-kernel summary of a `rocprofv3 --kernel-trace --stats` run (sqlite output) as CSV:
    python tools/kernel_stats.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, n, tot, avg, mn, mx in rows:
        w.writerow([name, n, tot, round(avg, 1), round(100.0 * tot / total, 3), mn, mx])
print(f"{len(rows)} kernels, {total / 1e6:.3f} ms of kernel time")
