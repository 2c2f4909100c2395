#!/bin/bash
# like ab_envs.sh with more rounds and steps: ms_per_step only.   tools/experiments/ab_envs_long.sh "" "VAR=1"
for i in 1 2 3 4; do
  for e in "$@"; do
    env WCT_DEBUG=1 $e python bench.py --steps 40 --warmup 5 --no-cpu-baseline --steps-only 2>/dev/null | TAG="${e:-default}" python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('%-44s ms_per_step %.3f' % (os.environ['TAG'], d['ms_per_step']))"
  done
done
