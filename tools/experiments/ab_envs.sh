#!/bin/bash
# A/B/C... of one build under several environments on ONE box, interleaved, two rounds:
#   FILTERS="enc_head dec_tail" tools/experiments/ab_envs.sh "" "VAR=1" "VAR=2 OTHER=3"
for i in 1 2; do
  for e in "$@"; do
    env WCT_DEBUG=1 $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --steps-only 2>/dev/null | TAG="${e:-default}" python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
ks={k['kernel']:k['ms_per_step'] for k in d['kernels']}
f=os.environ.get('FILTERS','').split()
sel=[k for k in ks if any(x in k for x in f)]
print('%-44s ms_per_step %.3f ' % (os.environ['TAG'], d['ms_per_step']) + '  '.join('%s=%.4f' % (k.replace('conv3x3_f16x3','c'), ks[k]) for k in sel))"
  done
done
