#!/bin/bash
# How much of a step is EXPOSED matrix-function time?  debug key eig_skip = N leaves the solves out after the first N (stale but
# identical results on repeated frames; --mode 16x only: the deferred outcome check of the wide models needs the solves to run):
# the drop in ms per step is what the solves cost the wall clock.  nscoop 1 / 0: single-launch / multi-launch C = 128 solves.
# usage (GPU box): tools/experiments/ab_eig_skip.sh   -> gpurun_out/ab_eig_skip.txt
OUT=gpurun_out/ab_eig_skip.txt
mkdir -p gpurun_out; : > $OUT
for r in 1 2; do
  for nc in 1 0; do
    for skip in 0 30; do
      ms=$(WCT_DEBUG=1 python bench.py --config cfg2 --steps 20 --warmup 3 --steps-only --no-cpu-baseline --debug-set nscoop=$nc --debug-set eig_skip=$skip 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "cfg2 round $r nscoop=$nc eig_skip=$skip ms_per_step=$ms" | tee -a $OUT
    done
  done
done
