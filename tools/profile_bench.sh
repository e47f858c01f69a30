#!/bin/bash
# Kernel trace + the two HBM traffic counter passes of bench.py, exported under gpurun_out/prof/.
#   bash tools/profile_bench.sh [tag]
set -e
tag=${1:-run}
root=$(pwd)
out=$root/gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="$root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-loop --no-train-step"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/kt_$tag -o kt -- python3 $B > $out/kt_$tag.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$tag -o pmc -- python3 $B > $out/fetch_$tag.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/write_$tag -o pmc -- python3 $B > $out/write_$tag.log 2>&1
cd $root
kt=$(find $out/kt_$tag -name '*.db' | head -n 1)
fe=$(find $out/fetch_$tag -name '*.db' | head -n 1)
wr=$(find $out/write_$tag -name '*.db' | head -n 1)
python3 tools/kernel_stats.py $kt $out/${tag}_kernel_stats.csv
python3 tools/collect_traffic.py $fe $wr $out/${tag}_traffic.json > $out/${tag}_traffic.txt
grep '^{' $out/kt_$tag.log > $out/${tag}_bench_under_trace.json || true
rm -rf $out/kt_$tag $out/fetch_$tag $out/write_$tag
# kernel trace of the training step (configs[4] shape, the "hip" backend): 1 warm-up + 3 timed steps + 1 for the loss
if [ -n "$PROFILE_TRAIN" ]; then
    cd /tmp
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/ktt_$tag -o kt -- python3 $root/tools/bench_next.py --only train --train-backends hip > $out/ktt_$tag.log 2>&1
    cd $root
    ktt=$(find $out/ktt_$tag -name '*.db' | head -n 1)
    python3 tools/kernel_stats.py $ktt $out/${tag}_train_kernel_stats.csv
    grep '^{' $out/ktt_$tag.log > $out/${tag}_train_under_trace.json || true
    rm -rf $out/ktt_$tag
fi
