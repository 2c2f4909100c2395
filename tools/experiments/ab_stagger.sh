#!/bin/bash
# A/B of wct_stylize's lane stagger (debug key "stagger"): interleaved rounds on one box, ms per step from the bench line.
# usage (GPU box): tools/experiments/ab_stagger.sh [rounds]    -> gpurun_out/ab_stagger.txt
R=${1:-3}
OUT=gpurun_out/ab_stagger.txt
mkdir -p gpurun_out; : > $OUT
for cfg in cfg2 cfg3; do
  for r in $(seq 1 $R); do
    for st in 1 0; do
      ms=$(python bench.py --config $cfg --steps 20 --warmup 3 --steps-only --no-cpu-baseline --debug-set stagger=$st 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "$cfg round $r stagger=$st ms_per_step=$ms" | tee -a $OUT
    done
  done
done
