#!/bin/bash
# A/B of build variants of the weight-stationary conv (libvdmi_ws_<name>.so, built by hand with -DWS_* switches) on one
# device: each variant runs tools/wsconv_check.py in its own process, twice, alternating.  Timing-only ablation builds
# fail the tool's accuracy check by construction; only the timing lines are kept.
out=${1:-gpurun_out/r2/variants.log}
: > "$out"
for round in 1 2; do
  for lib in video_diffusion_amd/libvdmi.so video_diffusion_amd/libvdmi_ws_*.so; do
    echo "== round $round $lib" >> "$out"
    VDMI_LIB_PATH=$lib timeout -k 10 120 python -u tools/wsconv_check.py --batch 16 --parts 1 --rounds 2 > /tmp/wsv.log 2>&1
    rc=$?
    if [ $rc -ge 124 ]; then echo "timeout/killed rc=$rc" >> "$out"; exit 1; fi
    grep "downs.0.1.block\|ups.2.1.block1\|sum over" /tmp/wsv.log >> "$out"
  done
done
for lib in video_diffusion_amd/libvdmi_wsd_*.so; do
  [ -e "$lib" ] || continue
  echo "== stamps $lib" >> "$out"
  VDMI_LIB_PATH=$lib timeout -k 10 120 python -u tools/ws_stamps.py --layer downs.0.1.block1 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
done
exit 0
