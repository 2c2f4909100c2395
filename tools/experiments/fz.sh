for f in 1 0; do echo "== WCT_FUSE=$f"; WCT_FUSE=$f python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])
for k in d['kernels']:
    if 'fused' in k['kernel'] or 'in3' in k['kernel'] or 'co=16' in k['kernel']: print('   ',k['kernel'],k['ms_per_step'],k['launches_per_step'])
"; done
