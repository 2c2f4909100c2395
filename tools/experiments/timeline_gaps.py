#!/usr/bin/env python3
"""Where an overlapped step's wall time goes: from a rocprofv3 kernel trace (tools/experiments/timeline.sh), for the LAST step,
 - wall = first start .. last end, per queue busy time,
 - time covered by at least one "big" kernel (grid >= 256 workgroups and >= 20 us) vs only small kernels vs nothing,
 - the list of the longest stretches without a big kernel and what ran in them.
usage: timeline_gaps.py <kernel_trace.csv> [steps]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"], grid // max(wg, 1)))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# step boundaries: a step starts with the same kernel sequence; use the number of launches per step
per = len([r for r in rows if not r[2].startswith(("void at::", "__amd"))]) // steps
rows = [r for r in rows if not r[2].startswith("void at::")]
last = rows[-per:]
t0, t1 = min(r[0] for r in last), max(r[1] for r in last)
print("launches in the last step: %d, wall %.3f ms" % (len(last), (t1 - t0) / 1e6))
byq = defaultdict(float)
for s, e, n, q, g in last:
    byq[q] += (e - s) / 1e6
print("busy per queue (ms):", dict((k, round(v, 3)) for k, v in byq.items()))
big = lambda r: r[4] >= 200 and (r[1] - r[0]) >= 15000   # noqa: E731


def union(iv):
    iv = sorted(iv)
    out, cs, ce = 0, None, None
    segs = []
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            segs.append((cs, ce)); cs, ce = s, e
    if cs is not None:
        segs.append((cs, ce))
    return segs


bigsegs = union([(r[0], r[1]) for r in last if big(r)])
anysegs = union([(r[0], r[1]) for r in last])
cov = lambda segs: sum(e - s for s, e in segs) / 1e6   # noqa: E731
print("covered by a big kernel: %.3f ms; by any kernel: %.3f ms; idle: %.3f ms" % (cov(bigsegs), cov(anysegs), (t1 - t0) / 1e6 - cov(anysegs)))
print("sum of big-kernel durations: %.3f ms (overlap between big kernels: %.3f)" % (sum(r[1] - r[0] for r in last if big(r)) / 1e6,
      sum(r[1] - r[0] for r in last if big(r)) / 1e6 - cov(bigsegs)))
gaps = []
prev = t0
for s, e in bigsegs + [(t1, t1)]:
    if s - prev > 20000:
        inside = defaultdict(lambda: [0, 0.0])
        for r in last:
            if r[0] < s and r[1] > prev and not big(r):
                k = inside[r[2][:60]]
                k[0] += 1; k[1] += (min(r[1], s) - max(r[0], prev)) / 1e3
        gaps.append((s - prev, prev - t0, dict(inside)))
    prev = max(prev, e)
gaps.sort(key=lambda g: -g[0])
print("stretches without a big kernel (> 20 us): %d, total %.3f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
for d, at, inside in gaps[:12]:
    top = sorted(inside.items(), key=lambda kv: -kv[1][1])[:4]
    print("  %.0f us at +%.3f ms: %s" % (d / 1e3, at / 1e6, "; ".join("%s x%d %.0fus" % (k, v[0], v[1]) for k, v in top)))
