#!/usr/bin/env python3
"""Scan the kernels' ISA for chains of adjacent dependent MFMAs (an MFMA whose accumulator input is the result of
the MFMA issued right before it, with at most `--gap` other instructions between them).

Why: a kernel that issued three such MFMAs back to back made a different kernel running on another HIP stream
return wrong values on the MI355X boxes this was developed on (the fold kernels of csrc/sla.hip and csrc/attn.hip,
since rewritten with separate accumulators).  Pairs separated by operand loads have not shown the effect; chains of
three or more are reported as an error.

    python tools/mfma_chain_scan.py [--gap 1] [file.hip ...]      (default: every .hip under video_diffusion_amd/csrc)
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video_diffusion_amd", "csrc")


def scan(asm_path, gap):
    worst = {}
    kern, prev, dist, run = None, None, 99, 1
    for line in open(asm_path):
        l = line.strip()
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, prev, dist, run = m.group(1), None, 99, 1
            continue
        if not l or l[0] in ";.":
            continue
        mm = re.match(r"v_mfma_\w+ ([av]\[\d+:\d+\]), [av]\[\d+:\d+\], [av]\[\d+:\d+\], (\S+)", l)
        if mm:
            dst, src_c = mm.groups()
            run = run + 1 if (prev is not None and dist <= gap and src_c == prev) else 1
            worst[kern] = max(worst.get(kern, 1), run)
            prev, dist = dst, 0
        else:
            dist += 1
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gap", type=int, default=1)
    ap.add_argument("files", nargs="*")
    args = ap.parse_args()
    files = args.files or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            out = os.path.join(tmp, os.path.basename(f) + ".s")
            # the flags of csrc/Makefile, per file (the attention kernels are built without SLP vectorisation)
            extra = ["-fno-slp-vectorize"] if os.path.basename(f).split(".")[0] in ("tattn", "tattnw", "sla", "attn", "xattn") else []
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, f, "-o", out] + extra,
                           check=True, stderr=subprocess.DEVNULL)
            for kern, n in sorted(scan(out, args.gap).items()):
                if n >= 2:
                    tag = "ERROR" if n >= 3 else "pair "
                    bad += n >= 3
                    print(f"{tag} {os.path.basename(f)}: {kern[:80]}: chain of {n} adjacent dependent MFMAs")
    print("chains of three or more:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
