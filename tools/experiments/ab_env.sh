#!/bin/bash
# A/B of one build under two environments on ONE box, interleaved: tools/experiments/ab_env.sh "VAR=1 OTHER=2" [kernel filters...]
ENVB=$1; shift
for i in 1 2; do
  for e in "" "$ENVB"; do
    env WCT_DEBUG=1 $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --steps-only 2>/dev/null | TAG="${e:-default}" python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
ks={k['kernel']:k['ms_per_step'] for k in d['kernels']}
sel=[k for k in ks if any(f in k for f in sys.argv[1:])] if len(sys.argv)>1 else []
print('%-24s ms_per_step %.3f ' % (os.environ['TAG'], d['ms_per_step']) + '  '.join('%s=%.4f' % (k.replace('conv3x3_f16x3','c'), ks[k]) for k in sel))" "$@"
  done
done
