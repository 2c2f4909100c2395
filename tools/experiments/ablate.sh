for ab in ${ABL:-0 16}; do echo "== WCT_ABLATE=$ab"; WCT_ABLATE=$ab python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('   step', d['ms_per_step'])
for k in d['kernels']:
    if k['kernel'].startswith('conv3x3_f16x3<co=') : print('   ',k['kernel'],k['ms_per_step'])
"; done
