for ab in ${ABL:-0 1 2 3 4 8 16 12 31}; do echo -n "== WCT_ABLATE=$ab  "; WCT_ABLATE=$ab python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
for k in d['kernels']:
    if k['kernel'].startswith('${KERN:-dec_tail}') : print(k['kernel'],k['ms_per_step'])
"; done
