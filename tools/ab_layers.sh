#!/bin/bash
# A/B of two builds of the library ON THE SAME GPU (devices of one pool differ by several percent, so numbers from
# separate runs do not compare):   tools/ab_layers.sh <a.so> <b.so> [rounds]
# Runs tools/layer_bench.py --all alternately with each library and prints the per-run sums; per-layer files are
# left in gpurun_out/ab_{a,b}<round>.txt.
set -e
A=$(readlink -f "$1"); B=$(readlink -f "$2"); N=${3:-2}
mkdir -p gpurun_out
for i in $(seq 1 "$N"); do
  VDMI_LIB_PATH=$A timeout -k 10 200 python tools/layer_bench.py --all > gpurun_out/ab_a$i.txt 2>&1
  VDMI_LIB_PATH=$B timeout -k 10 200 python tools/layer_bench.py --all > gpurun_out/ab_b$i.txt 2>&1
  echo "round $i  a: $(tail -n 1 gpurun_out/ab_a$i.txt)   b: $(tail -n 1 gpurun_out/ab_b$i.txt)"
done
