#!/usr/bin/env python3
"""Timeline of ONE denoise step of bench.py from a `rocprofv3 --kernel-trace` run (sqlite output): every kernel of
the step with start / end relative to the step's first kernel, its queue, and how many other kernels ran beside it.

    python tools/step_timeline.py <results.db> [step_index] > timeline.txt

A step ends with the posterior kernel (ddpm_step_kernel); step_index counts those (default: the 4th, inside the
timed graph replays of `bench.py --steps 5 --warmup 2`)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
pick = lambda *names: next(c for c in names if c in cols)
c_start, c_end, c_name = pick("start", "start_timestamp"), pick("end", "end_timestamp"), pick("name", "kernel_name")
c_q = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
rows = db.execute(f"select {c_start}, {c_end}, {c_name}, {c_q or 0} from kernels order by {c_start}").fetchall()
ends = [i for i, r in enumerate(rows) if "ddpm_step" in r[2]]
lo, hi = ends[which - 1] + 1, ends[which] + 1
step = rows[lo:hi]
t0 = step[0][0]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:46]


print(f"# columns of the kernels view: {cols}")
print(f"# step {which}: {len(step)} kernels, wall {(step[-1][1] - t0) / 1e3:.1f} us, sum of durations "
      f"{sum(r[1] - r[0] for r in step) / 1e3:.1f} us")
print(f"{'start_us':>9} {'end_us':>9} {'dur_us':>7} {'queue':>6} {'beside':>6}  kernel")
busy = 0.0
for s, e, n, q in step:
    beside = sum(1 for s2, e2, _, _ in step if s2 < e and e2 > s) - 1
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {str(q):>6} {beside:6d}  {short(n)}")
# time with 0 / 1 / 2+ kernels in flight
ev = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[min(depth, 2)] = hist.get(min(depth, 2), 0) + (t - last)
    depth += d
    last = t
print("# wall time by kernels in flight: " + ", ".join(f"{k}{'+' if k == 2 else ''}: {v / 1e3:.1f} us" for k, v in sorted(hist.items())))
