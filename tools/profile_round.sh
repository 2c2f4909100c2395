#!/bin/bash
# Profiling recipe of a round (run on the GPU box via gpurun); the summaries land in gpurun_out/prof_<tag>/ AND, named per round, in
# profiles/ (commit them from there):
#   1. rocprofv3 --kernel-trace --stats of the bench command (kernels one at a time: WCT_OVERLAP=0)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE -- separately, with --kernel-trace only) for HBM bytes per launch
#   3. one PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) for matrix-core utilisation per kernel
#   4. (SQ=1) three PMC passes of SQ issue / wait counters for the fused ends (tools/pmc_generic_summary.py)
# usage: tools/profile_round.sh <tag> [cfg2|cfg3]      -> gpurun_out/prof_<tag>/, profiles/<tag>_*
# cfg2 also refreshes profiles/hbm_traffic_latest.{txt,json} (with the source id bench.py compares against: `traffic_source.stale`)
set -e
TAG=${1:-rXX}
CFG=${2:-cfg2}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT profiles
export TMPDIR=/tmp
CMD="python $PWD/bench.py --config $CFG --steps 4 --warmup 1 --no-cpu-baseline --steps-only"
ROOT=$PWD
cd /tmp
WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats_bench.log 2> $OUT/stats_bench.err || echo "stats pass failed"
WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1 || echo "fetch pass failed"
WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1 || echo "write pass failed"
WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- $CMD > $OUT/pmc_mfma.log 2>&1 || echo "mfma pass failed"
if [ -n "$SQ" ]; then
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -o pmc -- $CMD > $OUT/sq$i.log 2>&1 || echo "sq pass $i failed"
  done
fi
cd $ROOT
python tools/pmc_summary.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/hbm_traffic.txt > /dev/null
python tools/mfma_summary.py $(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/mfma_util.txt > /dev/null
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
if [ -n "$SQ" ]; then
  python tools/pmc_generic_summary.py $(find $OUT/sq1 $OUT/sq2 $OUT/sq3 -name "*counter_collection.csv") --match enc_head,dec_tail,l1_,conv3x3_sp_kernel\<2 > $OUT/sq_counters.txt 2>&1 || true
fi
# keep the merge-back small: drop the raw per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
# the bench line of the stats pass (HIP-event durations of the same run as the rocprofv3 stats)
grep -h "^{" $OUT/stats_bench.log | tail -1 > $OUT/bench_line_under_rocprof.json || true
cp $OUT/kernel_stats.csv profiles/${TAG}_kernel_stats_no_overlap_steps_only.csv
cp $OUT/hbm_traffic.txt profiles/${TAG}_hbm_traffic_pmc.txt
cp $OUT/mfma_util.txt profiles/${TAG}_mfma_utilisation_pmc.txt
cp $OUT/bench_line_under_rocprof.json profiles/${TAG}_bench_line_under_rocprof_steps_only.json
[ -n "$SQ" ] && cp $OUT/sq_counters.txt profiles/${TAG}_sq_counters_fused_ends.txt
if [ "$CFG" = "cfg2" ]; then cp $OUT/hbm_traffic.txt profiles/hbm_traffic_latest.txt; cp $OUT/hbm_traffic.json profiles/hbm_traffic_latest.json; fi
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* profiles/hbm_traffic_latest.* gpurun_out/profiles_$TAG/ 2>/dev/null || true
ls -la $OUT | head -30
