#!/bin/bash
# VERDICT r4 task 1: the Winograd go / no-go measurement (tools/experiments/winograd_skeleton.hip) with constant and random operands,
# and the matrix-core-busy counters of the same launches.  usage (GPU box): tools/experiments/winograd_go_nogo.sh <tag> -> gpurun_out/winograd_<tag>/
TAG=${1:-r05}
REPO=$PWD
OUT=$REPO/gpurun_out/winograd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
$REPO/tools/experiments/winograd_skeleton > $OUT/skeleton_const.txt 2>&1
RANDOM_FILL=1 $REPO/tools/experiments/winograd_skeleton > $OUT/skeleton_random.txt 2>&1
RANDOM_FILL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- $REPO/tools/experiments/winograd_skeleton > $OUT/stats.log 2>&1 || echo "stats pass failed"
RANDOM_FILL=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- $REPO/tools/experiments/winograd_skeleton > $OUT/pmc.log 2>&1 || echo "pmc pass failed"
python $REPO/tools/mfma_summary.py $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $OUT/mfma_busy_random.txt --per-dispatch > /dev/null 2>&1 || true
RANDOM_FILL=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc2 -o pmc -- $REPO/tools/experiments/winograd_skeleton > $OUT/pmc2.log 2>&1 || echo "pmc2 pass failed"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python - <<PY > $OUT/lds_counters.txt 2>&1
import csv, glob, collections
f = glob.glob("$OUT/pmc2/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(f[0])):
    acc[row["Kernel_Name"][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items():
    print(k, dict(v))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/skeleton_const.txt $OUT/skeleton_random.txt $OUT/mfma_busy_random.txt $OUT/lds_counters.txt
