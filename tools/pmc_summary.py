#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from its sqlite output:
    python tools/pmc_summary.py <results.db> [kernel-name-substring] [last-N-dispatches]
When SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE were collected, a table follows: matrix-pipe utilisation =
busy SIMD-cycles / (GUI-active cycles summed over the 8 XCDs / 8 * 1024 SIMDs), and the share of LDS cycles lost
to bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # only the last N dispatches of the run
maxd = db.execute("select max(dispatch_id) from counters_collection").fetchone()[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(dict)
for name, cname, val, did, d in db.execute(
        "select kernel_name, counter_name, value, dispatch_id, duration from counters_collection"):
    if sub not in name or (last and did <= maxd - last):
        continue
    acc[name][cname] += val
    disp[name].add(did)
    dur[name][did] = d
for k, d in acc.items():
    n = len(disp[k])
    print(f"{k[:120]}  ({n} dispatches, mean {sum(dur[k].values()) / n / 1e3:.1f} us under the profiler)")
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v / n:16.1f} per dispatch")

rows = []
for k, d in acc.items():
    n = len(disp[k])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE"):
        util = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] * 128.0)
        lds = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"] if d.get("SQ_LDS_IDX_ACTIVE") else 0.0
        rows.append((sum(dur[k].values()), k, n, util, lds))
if rows:
    print("\nkernel | dispatches | mean us | MFMA busy / available | LDS cycles lost to bank conflicts")
    for tot, k, n, util, lds in sorted(rows, reverse=True):
        if util > 0 or lds > 0:
            print(f"{k[:100]} | {n} | {tot / n / 1e3:.1f} | {util:.3f} | {lds:.3f}")
