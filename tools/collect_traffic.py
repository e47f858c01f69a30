#!/usr/bin/env python3
"""HBM-side traffic per launch of the tap-GEMM kernel classes from two rocprofv3 --pmc passes
(FETCH_SIZE in one run, WRITE_SIZE in another; both are reported in KiB).

    python tools/collect_traffic.py <fetch_results.db> <write_results.db> profiles/traffic.json

On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Keys are bench.py's kernel-class names."""
import collections
import json
import re
import sqlite3
import sys


def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath)
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for name, cname, val, did in db.execute(
            "select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cname != counter:
            continue
        acc[name] += val
        n[name].add(did)
    return {k: acc[k] / len(n[k]) for k in acc}


def klass(name):
    m = re.search(r"tapgemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        return f"tapgemm<epi={m.group(6)},bn={m.group(2)}>"
    if "wsconv64_kernel" in name:
        return "wsconv64"
    for k in ("tattn_fused", "sla_pass1", "sla_pass2", "sla_fold", "rmsnorm_silu"):
        if k in name:
            return k
    return None


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for name in fetch:
    k = klass(name)
    if k is None:
        continue
    d = out.setdefault(k, {"fetch_bytes_per_launch": 0.0, "write_bytes_per_launch": 0.0, "kernels": []})
    # several tile shapes can share a class: keep the largest (it is the class's dominant shape)
    fb = fetch[name] * 1024 * 2
    wb = write.get(name, 0.0) * 1024
    if fb + wb > d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"]:
        d["fetch_bytes_per_launch"], d["write_bytes_per_launch"] = fb, wb
    d["kernels"].append(name)
res = {k: round(v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for k, v in out.items()}
res["_detail"] = out
res["_note"] = "bytes per launch, averaged over the launches of the class's kernel; FETCH_SIZE doubled (gfx950)"
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))
