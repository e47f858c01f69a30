#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace database (rocpd SQLite), per step.

    python tools/trace_summary.py gpurun_out/prof/x_results.db --steps 7 [--top 40] [--match tattn]
"""
import argparse
import collections
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=1, help="steps the traced run made (totals are divided by it)")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--match", default="")
    ap.add_argument("--csv", default="", help="also write every kernel's per-step totals to this file")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    agg = collections.defaultdict(lambda: [0, 0])
    for name, start, end in db.execute("select name, start, end from kernels"):
        name = name.replace("(anonymous namespace)::", "")
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*", "", name)
        agg[name][0] += 1
        agg[name][1] += end - start
    total = sum(v[1] for v in agg.values())
    print(f"kernel time per step: {total / 1e6 / a.steps:.3f} ms, launches per step: {sum(v[0] for v in agg.values()) / a.steps:.0f}")
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("kernel,launches_per_step,ms_per_step,avg_us\n")
            for name, (cnt, ns) in rows:
                f.write(f"\"{name}\",{cnt / a.steps:.2f},{ns / 1e6 / a.steps:.4f},{ns / cnt / 1e3:.2f}\n")
    if a.match:
        rows = [r for r in rows if a.match in r[0]]
    for name, (cnt, ns) in rows[:a.top]:
        print(f"{ns / 1e6 / a.steps:8.3f} ms {cnt / a.steps:7.1f} x {ns / cnt / 1e3:8.1f} us  {name[:110]}")


if __name__ == "__main__":
    main()
