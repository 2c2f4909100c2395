#!/bin/bash
# A/B of two builds of libwct_hip on ONE box, interleaved: tools/experiments/ab_lib.sh <other.so> [kernel name filters...]
OTHER=$1; shift
for i in 1 2; do
  for lib in "" "$OTHER"; do
    WCT_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --steps-only 2>/dev/null | LIBTAG=${lib:-current} python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
ks={k['kernel']:k['ms_per_step'] for k in d['kernels']}
sel=[k for k in ks if any(f in k for f in sys.argv[1:])] if len(sys.argv)>1 else []
print('%-28s ms_per_step %.3f ' % (os.path.basename(os.environ['LIBTAG']), d['ms_per_step']) + '  '.join('%s=%.4f' % (k.split('<')[0], ks[k]) for k in sel))" "$@"
  done
done
