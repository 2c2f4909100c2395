#!/bin/bash
# Do the persistent convolution kernels (one workgroup per CU, all of its LDS and registers) starve the other lane's small
# matrix-function launches?  WCT_CU_RESERVE = k sizes the persistent grids for (CUs - k), with multi-launch (nscoop 0) and
# single-launch (1) solves.  Round 3: reserve 8 / 16 cost 0.1 ms per 4K step with the multi-launch solves (the CUs given up
# outweigh the quicker launches); the single-launch solve is what got the time back.
# usage (GPU box): tools/experiments/ab_cu_reserve.sh [cfg] -> gpurun_out/ab_cu_reserve_<cfg>.txt
CFG=${1:-cfg2}
OUT=gpurun_out/ab_cu_reserve_$CFG.txt
mkdir -p gpurun_out; : > $OUT
for r in 1 2; do
  for res in 0 8 16; do
    for nc in 0 1; do
      ms=$(WCT_CU_RESERVE=$res python bench.py --config $CFG --steps 20 --warmup 3 --steps-only --no-cpu-baseline --debug-set nscoop=$nc 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "$CFG round $r reserve=$res nscoop=$nc ms_per_step=$ms" | tee -a $OUT
    done
  done
done
