#!/bin/bash
# How much of a step is EXPOSED matrix-function time?  debug key eig_skip = N leaves the solves out after the first N (stale but
# identical results on repeated frames): the drop in ms per step is what the solves cost the wall clock, stagger on and off.
# usage (GPU box): tools/experiments/ab_eig_skip.sh   -> gpurun_out/ab_eig_skip.txt
OUT=gpurun_out/ab_eig_skip.txt
mkdir -p gpurun_out; : > $OUT
for cfg in cfg2 cfg3; do
  for r in 1 2; do
    for st in 1 0; do
      for skip in 0 30; do
        ms=$(python bench.py --config $cfg --steps 20 --warmup 3 --steps-only --no-cpu-baseline --debug-set stagger=$st --debug-set eig_skip=$skip 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
        echo "$cfg round $r stagger=$st eig_skip=$skip ms_per_step=$ms" | tee -a $OUT
      done
    done
  done
done
