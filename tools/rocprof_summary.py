#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite database) as per-kernel statistics.
usage: tools/rocprof_summary.py <results.db> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["# rocprofv3 --kernel-trace --stats summary (from %s)" % sys.argv[1].split("/")[-1],
             "%-112s %7s %12s %11s %11s %11s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%")]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-112s %7d %12.3f %11.2f %11.2f %11.2f %6.2f" % (short(n), a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
    lines.append("TOTAL kernel time %.3f ms over %d dispatches" % (tot / 1e6, len(rows)))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
