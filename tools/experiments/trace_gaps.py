"""From a rocprofv3 kernel trace of cached_loop.py: the last frame's timeline -- per kernel start/duration and the idle gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame = after the last l1_decode-before-last ... take the last 1/4 of kernels roughly: find the last 'enc_head' sequence start
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "l1_decode" in n]
start = idx[-2] + 1 if len(idx) >= 2 else 0
sel = rows[start:idx[-1] + 1]
t0 = int(sel[0]["Start_Timestamp"])
busy = 0; gap_total = 0; prev_end = t0
out = []
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    gap_total += max(gap, 0); busy += e - s
    short = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    out.append((s - t0, e - s, gap, short))
    prev_end = max(prev_end, e)
print("frame: %.3f ms wall, kernels %.3f ms, gaps %.3f ms, %d launches" % ((prev_end - t0) / 1e6, busy / 1e6, gap_total / 1e6, len(sel)))
# aggregate small-kernel stretches: consecutive kernels shorter than 40 us
i = 0
while i < len(out):
    if out[i][1] < 40000:
        j = i
        while j < len(out) and out[j][1] < 40000: j += 1
        if j - i >= 5:
            span = out[j - 1][0] + out[j - 1][1] - out[i][0]
            print("  stretch of %3d short kernels at %.3f ms: %.3f ms (first %s)" % (j - i, out[i][0] / 1e6, span / 1e6, out[i][3]))
        i = j
    else:
        i += 1
