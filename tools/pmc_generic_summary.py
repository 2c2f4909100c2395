#!/usr/bin/env python3
"""Per-kernel averages of whatever counters one or more rocprofv3 --pmc passes hold: tools/pmc_generic_summary.py a.csv [b.csv ...] [--match substr,...]
(one row per kernel, one column per counter: value per launch; rocprofv3 sums a counter over XCDs / SEs.)"""
import csv, re, sys
from collections import defaultdict
files = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1].split(",")
    files = [f for f in files if f != sys.argv[sys.argv.index("--match") + 1]]
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(set)); dur = defaultdict(float); dn = defaultdict(set)
names = []
for fi, f in enumerate(files):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if match and not any(m in k for m in match):
            continue
        c = r["Counter_Name"]
        if c not in names: names.append(c)
        agg[k][c] += float(r["Counter_Value"]); calls[k][c].add((fi, r["Dispatch_Id"]))
        if (fi, r["Dispatch_Id"]) not in dn[k]:
            dn[k].add((fi, r["Dispatch_Id"])); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%-44s %8s " % ("kernel", "avg_us") + " ".join("%22s" % n[-22:] for n in names))
for k in sorted(agg, key=lambda k: -dur[k]):
    print("%-44s %8.1f " % (k[:44], dur[k] / len(dn[k])) + " ".join("%22.0f" % (agg[k][n] / max(len(calls[k][n]), 1)) for n in names))
