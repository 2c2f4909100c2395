#!/bin/bash
# A/B/C... of several builds of libwct_hip on ONE box, interleaved, two rounds:
#   tools/experiments/ab_libs.sh "lib1.so lib2.so ..." [kernel name filters...]     ("" = the in-tree build)
LIBS=$1; shift
for i in 1 2; do
  for lib in "" $LIBS; do
    WCT_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --steps-only 2>/dev/null | LIBTAG=${lib:-current} python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
ks={k['kernel']:k['ms_per_step'] for k in d['kernels']}
sel=[k for k in ks if any(f in k for f in sys.argv[1:])] if len(sys.argv)>1 else []
print('%-24s ms_per_step %.3f ' % (os.path.basename(os.environ['LIBTAG']), d['ms_per_step']) + '  '.join('%s=%.4f' % (k.replace('conv3x3_f16x3','c'), ks[k]) for k in sel))" "$@"
  done
done
