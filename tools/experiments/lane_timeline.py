#!/usr/bin/env python3
"""What the two lanes do during ONE overlapped stylise call: HIP-event timestamps of every profile scope (WCT_TIMELINE, wct_api.hip
prof_collect), printed as a merged timeline plus: wall, per-lane busy time, time with both / one / no lane inside a scope, and the
time in which ONLY matrix functions (or other tiny launches) were running.
usage (GPU box): python tools/experiments/lane_timeline.py [cfg2|cfg3] [interleave 1|0]  -> gpurun_out/lane_timeline_<cfg>_<interleave>.txt"""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "collaborative-distillation_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
interleave = int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
raw = os.path.join(REPO, "gpurun_out", "lane_timeline_%s_%d.raw" % (cfg, interleave))
if os.path.exists(raw):
    os.remove(raw)
os.environ["WCT_DEBUG"] = "1"
os.environ["WCT_TIMELINE"] = raw
import torch  # noqa: E402
from fixture_compare import cfg2_frames, cfg3_frames  # noqa: E402
from wct_hip import WCT, model_zoo  # noqa: E402

if cfg == "cfg3":
    w = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=model_zoo.synth_weights("original", 3))
    c, s = cfg3_frames()
else:
    w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz")))
    c, s = cfg2_frames()
w.debug_set("interleave", interleave)
c, s = torch.from_numpy(c).cuda(), torch.from_numpy(s).cuda()
out = torch.empty_like(c)
for _ in range(3):
    w.stylize(c, s, out=out)
torch.cuda.synchronize()
w.profile_reset(); w.profile(True)
w.stylize(c, s, out=out)
torch.cuda.synchronize()
w.profile_read()
w.profile(False)

recs = []
for ln in open(raw):
    if ln.startswith("#"):
        continue
    name, lane, t0, t1 = ln.rsplit(None, 3)
    recs.append((float(t0), float(t1), lane, name))
recs.sort()
small = lambda n: n.startswith(("matfun", "fold", "assemble", "split"))  # noqa: E731
T0, T1 = min(r[0] for r in recs), max(r[1] for r in recs)
lines = ["# %s interleave=%d: %d scopes, wall %.3f ms (ONE cold call: synchronised before it)" % (cfg, interleave, len(recs), T1 - T0)]
# sweep
pts = sorted({r[0] for r in recs} | {r[1] for r in recs})
acc = {"both_big": 0.0, "one_big": 0.0, "only_small": 0.0, "idle": 0.0, "big+small": 0.0}
for a, b in zip(pts[:-1], pts[1:]):
    act = [r for r in recs if r[0] <= a and r[1] >= b]
    nb = sum(1 for r in act if not small(r[3]))
    ns = sum(1 for r in act if small(r[3]))
    key = "both_big" if nb >= 2 else ("big+small" if nb == 1 and ns else ("one_big" if nb == 1 else ("only_small" if ns else "idle")))
    acc[key] += b - a
lines.append("# time (ms) with: " + ", ".join("%s %.3f" % kv for kv in acc.items()))
for lane in ("main", "side"):
    lines.append("# %s lane: %.3f ms inside scopes (%.3f in matrix functions / folds)" % (
        lane, sum(r[1] - r[0] for r in recs if r[2] == lane), sum(r[1] - r[0] for r in recs if r[2] == lane and small(r[3]))))
for t0, t1, lane, name in recs:
    lines.append("%8.3f %8.3f %7.3f  %s%s" % (t0 - T0, t1 - T0, t1 - t0, "" if lane == "main" else " " * 40, name))
open(os.path.join(REPO, "gpurun_out", "lane_timeline_%s_%d.txt" % (cfg, interleave)), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:4]))
