#!/bin/bash
# which launches surround the runtime's copy kernels inside a stylise step?  tools/experiments/trace_copies.sh
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/trace; mkdir -p $OUT
CMD="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --steps-only"
cd /tmp; WCT_DEBUG=1 WCT_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- $CMD > $OUT/log.txt 2>&1; cd - > /dev/null
python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:50] for r in rows]
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print(len(rows), "launches,", len(idx), "copyBuffer")
for i in idx[-20:]:
    print(" | ".join(names[max(0, i - 2):i + 2]), "| grid", rows[i].get("Grid_Size"), "dur", int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]))
PY
rm -rf gpurun_out/trace
