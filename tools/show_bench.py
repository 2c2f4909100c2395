"""Print a bench.py JSON line (file or stdin) as a per-kernel table."""
import json, sys
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for line in src:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("step %.3f ms  %s %.1f %s  n_gpus=%d" % (d["ms_per_step"], d["metric"][:30], d["value"], d["unit"], d["n_gpus"]))
    tot = 0.0
    for k in d.get("kernels") or []:
        tot += k["ms_per_step"]
        print("  %-34s %7.3f ms  x%-3d %8s TF  %8s GB/s" % (k["kernel"], k["ms_per_step"], k["launches_per_step"], k.get("tflops"), k.get("algo_GBs")))
    print("  kernel sum %.3f ms" % tot)
    for key in ("roofline", "passes", "cpu_baseline"):
        if d.get(key):
            print(" ", key, json.dumps(d[key]))
