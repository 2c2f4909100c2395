#!/bin/bash
# Do the persistent convolution kernels (one workgroup per CU, all of its LDS and registers) starve the other lane's small
# matrix-function launches?  WCT_CU_RESERVE = k sizes the persistent grids for (CUs - k); with and without the lane stagger.
# usage (GPU box): tools/experiments/ab_cu_reserve.sh [cfg] -> gpurun_out/ab_cu_reserve.txt
CFG=${1:-cfg2}
OUT=gpurun_out/ab_cu_reserve_$CFG.txt
mkdir -p gpurun_out; : > $OUT
for r in 1 2; do
  for res in 0 8 16; do
    for st in 0 1; do
      ms=$(WCT_CU_RESERVE=$res python bench.py --config $CFG --steps 20 --warmup 3 --steps-only --no-cpu-baseline --debug-set stagger=$st 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "$CFG round $r reserve=$res stagger=$st ms_per_step=$ms" | tee -a $OUT
    done
  done
done
