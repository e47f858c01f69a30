#!/bin/bash
# A/B of an environment switch on ONE GPU (devices differ by several percent): alternate bench.py runs with
# `VAR=a` and `VAR=b` and print ms/step.   tools/ab_env.sh VAR a b [rounds]
set -e
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq 1 "$N"); do
  for v in "$A" "$B"; do
    ms=$(env "$VAR=$v" timeout -k 10 200 python bench.py --no-cpu-baseline --no-full-loop --steps 40 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $i  $VAR=$v  $ms ms/step"
  done
done
