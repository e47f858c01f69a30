#!/bin/bash
# A/B of two environment settings on ONE GPU: alternate bench.py runs under `env $A` and `env $B` and print ms/step.
#   tools/ab_cmd.sh "VAR1=x VAR2=y" "VAR1=z" [rounds] [steps]
set -e
A=$1; B=$2; N=${3:-3}; S=${4:-40}
for i in $(seq 1 "$N"); do
  for v in "$A" "$B"; do
    ms=$(env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-full-loop --no-train-step --steps "$S" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $i  [$v]  $ms ms/step"
  done
done
