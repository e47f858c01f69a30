#!/bin/bash
# One SQ counter pass over bench.py (MFMA busy cycles, LDS activity / bank conflicts), summarised per kernel.
#   bash tools/profile_pmc.sh [tag]
set -e
tag=${1:-run}
root=$(pwd)
out=$root/gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    -d $out/sq_$tag -o pmc -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-loop > $out/sq_$tag.log 2>&1
cd $root
db=$(find $out/sq_$tag -name '*.db' | head -n 1)
python3 tools/pmc_summary.py $db > $out/${tag}_pmc_sq.txt
rm -rf $out/sq_$tag
