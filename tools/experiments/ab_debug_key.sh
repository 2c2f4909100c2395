#!/bin/bash
# Same-box, interleaved A/B of one context debug key (wct_debug_set) on the timed loop: ms_per_step of `bench.py --steps-only` and the
# kernel families that moved.   usage (GPU box): tools/experiments/ab_debug_key.sh KEY "V1 V2 ..." [cfg2|cfg3] [rounds]
#   -> gpurun_out/ab_<KEY>_<cfg>.txt
KEY=$1; VALS=${2:-"0 1"}; CFG=${3:-cfg2}; ROUNDS=${4:-3}
OUT=gpurun_out/ab_${KEY}_${CFG}.txt
mkdir -p gpurun_out; : > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    python bench.py --config $CFG --steps 20 --warmup 3 --steps-only --no-cpu-baseline --debug-set $KEY=$v 2>/dev/null > /tmp/ab_line.json
    python - "$KEY" "$v" "$r" "$CFG" <<'PY' | tee -a $OUT
import json, sys
d = json.load(open("/tmp/ab_line.json"))
k = {x["kernel"]: x["ms_per_step"] for x in d.get("kernels", [])}
pick = [n for n in k if any(t in n for t in ("moments", "matfun", "in3", "enc_head"))]
print("%s round %s %s=%s ms_per_step=%.3f  %s" % (sys.argv[4], sys.argv[3], sys.argv[1], sys.argv[2], d["ms_per_step"], "  ".join("%s=%.3f" % (n, k[n]) for n in sorted(pick))))
PY
  done
done
