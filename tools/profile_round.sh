#!/bin/bash
# Profiling recipe of a round (run on the GPU box via gpurun; summaries are copied into profiles/ by hand afterwards):
#   1. rocprofv3 --kernel-trace --stats of the bench command (kernels one at a time: WCT_OVERLAP=0)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE -- separately, with --kernel-trace only) for HBM bytes per launch
#   3. one PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) for matrix-core utilisation per kernel
# usage: tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --steps-only"
cd /tmp
WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats_bench.log 2>&1 || echo "stats pass failed"
WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1 || echo "fetch pass failed"
WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1 || echo "write pass failed"
WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- $CMD > $OUT/pmc_mfma.log 2>&1 || echo "mfma pass failed"
cd - > /dev/null
find $OUT -name "*.csv" | head -20
python tools/pmc_summary.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/hbm_traffic.txt > /dev/null
python tools/mfma_summary.py $(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/mfma_util.txt > /dev/null
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep the merge-back small: drop the raw per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT
