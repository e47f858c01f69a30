#!/usr/bin/env python3
"""Launches of the kernels whose name contains `match` in a rocprofv3 --kernel-trace database, grouped by grid shape.

    python tools/trace_by_grid.py <results.db> <match> [steps]
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
match = sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
agg = collections.defaultdict(lambda: [0, 0])
for name, s, e, gx, gy, gz, wx in db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels"):
    if match in name:
        k = (gx // max(wx, 1), gy, gz)
        agg[k][0] += 1
        agg[k][1] += e - s
print(f"{'workgroups (x, y, z)':>28} {'per step':>9} {'avg us':>8} {'ms/step':>8}")
for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(k):>28} {n / steps:9.1f} {ns / n / 1e3:8.1f} {ns / 1e6 / steps:8.3f}")
