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
    """{kernel name: (sum of the counter over its dispatches, number of dispatches)}"""
    db = sqlite3.connect(dbpath)
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for name, cname, val, did in db.execute(
            "select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cname != counter:
            continue
        acc[name] += val
        n[name].add(did)
    return {k: (acc[k], len(n[k])) for k in acc}


def klass(name):
    m = re.search(r"tapgemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, \w+)*>", name)
    if m:
        return f"tapgemm<epi={m.group(6)},bn={m.group(2)}>"
    if "wsconv64_kernel" in name:
        return "wsconv64"
    m = re.search(r"dconv_kernel<\d+, \d+, (true|false)>", name)
    if m:
        return "dconv<epi=1>" if m.group(1) == "true" else "dconv"      # the RMSNorm-epilogue form is a class of its own
    m = re.search(r"tattn_wide_kernel<\d+, \w+, (\d)>", name)
    if m:
        return ("tattn_fused", "sattn_fused", "sla_fused_wide")[int(m.group(1))]
    if "sla_wide_long_kernel" in name:
        return "sla_fused_wide"
    for k in ("tattn_fused", "sla_pass1", "sla_pass2", "sla_fold", "rmsnorm_silu", "ddpm_step", "attn_core_lds",
              "xattn_core", "init_conv_mfma", "final_conv"):
        if k in name:
            return k
    return None


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for name, (fsum, fn) in fetch.items():
    k = klass(name)
    if k is None:
        continue
    wsum, wn = write.get(name, (0.0, fn))
    d = out.setdefault(k, {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches": 0, "kernels": {}})
    fb, wb = fsum * 1024 * 2, wsum * 1024 * (fn / wn if wn else 0.0)     # write pass scaled to the fetch pass's dispatch count
    d["fetch_bytes"] += fb
    d["write_bytes"] += wb
    d["launches"] += fn
    d["kernels"][name] = {"launches": fn, "fetch_bytes_per_launch": fb / fn, "write_bytes_per_launch": wb / fn}
# per class: bytes per launch averaged over every launch of the class in the profiled run (all its kernels and shapes)
res = {k: round((v["fetch_bytes"] + v["write_bytes"]) / v["launches"]) for k, v in out.items()}
res["_detail"] = {k: {"fetch_bytes_per_launch": v["fetch_bytes"] / v["launches"],
                      "write_bytes_per_launch": v["write_bytes"] / v["launches"], "kernels": v["kernels"]}
                  for k, v in out.items()}
res["_note"] = ("bytes per launch, averaged over all launches of the class (every kernel and shape of it) in the profiled "
                "bench run; FETCH_SIZE doubled (gfx950)")
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))
