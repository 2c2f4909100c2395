WCT_PROF_SHAPES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --steps-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'])
for k in d['kernels']:
    print('%-58s %8.4f ms %3d launches  %7s TF  %8s GB/s' % (k['kernel'], k['ms_per_step'], k['launches_per_step'], k['tflops'], k['algo_GBs']))
"
