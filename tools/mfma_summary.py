#!/usr/bin/env python3
"""Per-kernel matrix-core utilisation from one rocprofv3 --pmc pass (CSV): SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE.
SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip, the cycles a SIMD's matrix core is busy (MI355X_MICROARCH.md: 32 per
32x32x16 f16 MFMA).  Utilisation = busy cycles / (1024 SIMDs x the kernel's duration in shader cycles); the duration in cycles is
GRBM_GUI_ACTIVE when that counter is in the pass (per-XCD values are summed by rocprofv3 -> / 8), else avg_us x --ghz.
usage: tools/mfma_summary.py <counter_collection.csv> [out.txt] [--ghz 2.0] [--per-dispatch]
--per-dispatch: one row per launch, in launch order (micro-benchmarks that launch one template with different arguments)"""
import csv
import re
import sys
from collections import defaultdict

SIMDS, XCDS = 1024, 8


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ghz = 2.0
    if "--ghz" in sys.argv:
        ghz = float(sys.argv[sys.argv.index("--ghz") + 1])
        args = [a for a in args if a != str(ghz) and a != sys.argv[sys.argv.index("--ghz") + 1]]
    per = "--per-dispatch" in sys.argv
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    dur = defaultdict(float)
    for r in csv.DictReader(open(args[0])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if per:
            k = "#%04d %s" % (int(r["Dispatch_Id"]), k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in calls[k]:
            calls[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    lines = ["# matrix-core utilisation per kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES [GRBM_GUI_ACTIVE]); %d SIMDs" % SIMDS,
             "%-64s %6s %9s %16s %14s %9s %8s" % ("kernel", "calls", "avg_us", "mfma_busy/call", "gui_active/call", "eff_GHz", "mfma_%")]
    rows = []
    for k, c in agg.items():
        n = len(calls[k])
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / n
        us = dur[k] / n
        cyc = gui / XCDS if gui else us * 1e3 * ghz
        rows.append((dur[k], k, n, us, busy, gui, (cyc / (us * 1e3)) if us else 0.0, 100.0 * busy / (SIMDS * cyc) if cyc else 0.0))
    for _, k, n, us, busy, gui, eff, util in (sorted(rows, key=lambda t: t[1]) if per else sorted(rows, reverse=True)):
        if k.startswith(("void at::", "__amd")):
            continue
        lines.append("%-64s %6d %9.1f %16.0f %14.0f %9.2f %8.1f" % (k[:64], n, us, busy, gui, eff, util))
    out = "\n".join(lines) + "\n"
    if len(args) > 1:
        open(args[1], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
