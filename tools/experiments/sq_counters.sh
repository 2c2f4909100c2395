#!/bin/bash
# SQ-level counters of the fused ends (issue / wait / LDS / VALU / MFMA), three PMC passes: tools/experiments/sq_counters.sh
set -e
OUT=$PWD/gpurun_out/sq; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --steps-only"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd - > /dev/null
python tools/pmc_generic_summary.py $(find $OUT -name "*counter_collection.csv") --match enc_head,dec_tail,l1_,conv3x3_sp_kernel\<2 > $OUT/summary.txt 2>&1 || true
find $OUT -name "*.csv" -delete
cat $OUT/summary.txt
