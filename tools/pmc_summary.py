#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output).
Units/corrections per MI355X_MICROARCH.md "HBM": the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B) -> doubled here; WRITE_SIZE as is.
usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.txt]"""
import csv
import re
import sys
from collections import defaultdict


def load(path, name):
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        a = agg[k]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    lines = ["# HBM traffic per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes); FETCH doubled (gfx950)",
             "%-70s %6s %12s %12s %12s %10s %9s" % ("kernel", "calls", "read_MB/call", "write_MB/call", "total_MB/call", "avg_us", "TB/s")]
    rows = []
    for k in f:
        n = f[k][0]
        rd = 2.0 * f[k][1] * 1024 / n / 1e6
        wr = w[k][1] * 1024 / max(w[k][0], 1) / 1e6 if k in w else 0.0
        us = f[k][2] / n
        rows.append((f[k][2], k, n, rd, wr, us))
    for _, k, n, rd, wr, us in sorted(rows, reverse=True):
        lines.append("%-70s %6d %12.2f %12.2f %12.2f %10.1f %9.2f" % (k[:70], n, rd, wr, rd + wr, us, (rd + wr) / us if us else 0))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out)
        import json
        js = {k: {"calls": n, "read_MB_per_launch": round(rd, 3), "write_MB_per_launch": round(wr, 3), "avg_us": round(us, 2)}
              for _, k, n, rd, wr, us in rows if not k.startswith(("void at::", "__amd"))}
        import hashlib, os
        h = hashlib.sha256()     # the same id bench.py computes: ties this summary to the kernel sources it was taken from
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "collaborative-distillation_amd", "csrc")
        for fn in sorted(os.listdir(d)):
            if fn.startswith(("conv", "level1", "moments", "wct_common")) and fn.endswith((".hip", ".h")):
                h.update(fn.encode())
                h.update(open(os.path.join(d, fn), "rb").read())
        json.dump({"source_id": h.hexdigest()[:16], "note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB units, "
                           "FETCH doubled per MI355X_MICROARCH.md); bench.py reports the entry of its dominant kernel as roofline.traffic",
                   "kernels": js}, open(sys.argv[3].rsplit(".", 1)[0] + ".json", "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
