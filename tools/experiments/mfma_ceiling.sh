#!/bin/bash
# the 64-cout convolution's skeleton (tools/experiments/mfma_loop.hip: the kernel's inner loop rebuilt from the
# tap loop outwards) under the SAME counters as the real kernel -- SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE -- with constant and
# with random operands: is the gap to the 2.5 PF peak idle matrix cores (schedule) or clock (DVFS)?
# usage (GPU box): tools/experiments/mfma_ceiling.sh <tag>   -> gpurun_out/mfma_ceiling_<tag>/
set -e
TAG=${1:-rXX}
REPO=$PWD
OUT=$REPO/gpurun_out/mfma_ceiling_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for mode in const random; do
  if [ $mode = random ]; then export RANDOM_FILL=1; fi
  $REPO/tools/experiments/mfma_loop > $OUT/loop_$mode.txt 2>&1 || true
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_$mode -o pmc -- $REPO/tools/experiments/mfma_loop > $OUT/pmc_$mode.log 2>&1 || echo "pmc pass failed ($mode)"
  python $REPO/tools/mfma_summary.py $(find $OUT/pmc_$mode -name "*counter_collection.csv" | head -1) $OUT/mfma_$mode.txt --per-dispatch > /dev/null || true
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
cat $OUT/loop_const.txt $OUT/mfma_const.txt $OUT/loop_random.txt $OUT/mfma_random.txt
