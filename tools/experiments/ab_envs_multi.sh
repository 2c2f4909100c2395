#!/bin/bash
# interleaved A/B/C... of ONE build under several values of one environment variable: tools/experiments/ab_envs_multi.sh VAR "v1 v2 ..." [kernel filters...]
VAR=$1; VALS=$2; shift; shift
for i in 1 2; do
  for v in $VALS; do
    env WCT_DEBUG=1 $VAR=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --steps-only 2>/dev/null | TAG="$VAR=$v" python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
ks={k['kernel']:k['ms_per_step'] for k in d['kernels']}
sel=[k for k in ks if any(f in k for f in sys.argv[1:])] if len(sys.argv)>1 else []
print('%-24s ms_per_step %.3f ' % (os.environ['TAG'], d['ms_per_step']) + '  '.join('%s=%.4f' % (k.replace('conv3x3_f16x3','c'), ks[k]) for k in sel))" "$@"
  done
done
