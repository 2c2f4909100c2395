"""Instruction budget of a kernel from its ISA listing (hipcc -S --cuda-device-only): the instructions of its LARGEST loop body (the persistent
tile loop of the fused kernels -- everything between the loop's head label and its backward branch, inner loops included once), by class:
MFMA, other VALU, LDS (ds_*), global / buffer memory, SALU (s_* except waits / nops / branches / barriers), waits + nops, branches, barriers.
    python tools/isa_budget.py listing.s name_fragment [name_fragment ...]
A static count: both arms of a uniform branch inside the loop (interior / image-border tiles) are counted, so it is an UPPER bound of what a wave
issues per tile; the dynamic counters (profiles/*_sq_counters_fused_ends.txt: SQ_INSTS_* / waves / tiles) are the measured figure beside it."""
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op in ("s_waitcnt", "s_nop") or op.startswith("s_waitcnt") or op.startswith("s_sleep"):
        return "wait_nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(path):
    cur, out = None, {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
            continue
        out[cur].append(ln.rstrip("\n"))
    return out


def largest_loop(lines):
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    best = (0, 0)
    for i, ln in enumerate(lines):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[1] - best[0]:
            best = (labels[m.group(1)], i)
    return best


def count(lines):
    c = {}
    for ln in lines:
        m = re.match(r"^\s+([a-z_0-9]+)\b", ln)
        if not m or ln.strip().startswith((".", ";")):
            continue
        k = classify(m.group(1))
        c[k] = c.get(k, 0) + 1
    return c


if __name__ == "__main__":
    ks = kernels(sys.argv[1])
    for frag in sys.argv[2:]:
        for name, lines in ks.items():
            if frag not in name:
                continue
            a, b = largest_loop(lines)
            c = count(lines[a:b + 1])
            tot = count(lines)
            order = ("mfma", "valu", "lds", "vmem", "salu", "wait_nop", "branch", "barrier")
            print("%s\n  tile loop (%d lines): %s\n  whole kernel:        %s" % (
                name[:110], b - a + 1, "  ".join("%s %d" % (k, c.get(k, 0)) for k in order), "  ".join("%s %d" % (k, tot.get(k, 0)) for k in order)))
