#!/bin/bash
# A/B on one box: style-side stream at default vs lowest priority (wct_debug_set side_priority), interleaved runs
for i in 1 2; do
  for v in 0 1; do
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --steps-only --debug-set side_priority=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('side_priority=$v  ms_per_step', d['ms_per_step'], 'MP/s', d.get('value') or d.get('unverified_value'))"
  done
done
