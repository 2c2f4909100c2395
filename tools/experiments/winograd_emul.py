"""CPU only, zero GPU minutes (round 4, VERDICT r3 task 3): would Winograd F(2x2, 3x3) with split-f16 products on the TRANSFORMED
tiles pass the reference-anchored gate?  The oracle's cascade is run with oracle/conv_emul.c's emulation in the layers a Winograd
kernel would take (cin >= CMIN and cout >= CMIN, not behind an upsample: those run as per-parity 2x2 convolutions already), every
other layer stays the oracle's fp32 convolution, and the result is compared with THE REFERENCE'S OWN PIXELS (G13 config 2, G14
config 3).  Arms: oracle (no emulation), direct (nine taps, f16x3: what the product's kernels do today), winograd (f16x3 on U, V),
winograd32 (fp32 operands: the algorithm's own error without the split).

usage: python tools/experiments/winograd_emul.py cfg2:noise|cfg2:smooth|cfg3 [arms...] [--cmin 64]
"""
import os, sys, time
REPO = "/root/repo"; sys.path[:0] = [REPO, REPO + "/collaborative-distillation_amd"]
import numpy as np
from oracle import wct_oracle
from wct_hip import model_zoo
from tests.fixture_compare import cfg2_frames, cfg3_frames, compare_to_fixture

args = [a for a in sys.argv[1:]]
cmin = 64
if "--cmin" in args:
    i = args.index("--cmin"); cmin = int(args[i + 1]); del args[i:i + 2]
frame = args[0]
arms = args[1:] or ["oracle", "direct", "winograd"]
wct_oracle.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
if frame.startswith("cfg2"):
    kind = frame.split(":")[1]
    mode, w = "16x", model_zoo.load_npz_weights(REPO + "/collaborative-distillation_amd/weights/16x.npz")
    c, s = cfg2_frames(kind)
    g = dict(np.load(REPO + "/tests/golden/g13_cfg2_%s.npz" % kind))
else:
    mode, w = "original", model_zoo.synth_weights("original", 3)
    c, s = cfg3_frames()
    g = dict(np.load(REPO + "/tests/golden/g14_cfg3_original.npz"))

count = {}


def hook_for(arm):
    if arm == "oracle":
        return None
    algo, split = {"direct": ("direct", True), "winograd": ("winograd", True), "winograd32": ("winograd", False)}[arm]

    def hook(kind, level, l, prev, x, wt, bs):
        if l.cin < cmin or l.cout < cmin or (kind == "dec" and prev is not None and prev.up_after):
            return None
        count[arm] = count.get(arm, 0) + 1
        return wct_oracle.conv3x3_emul(x, wt, bs, True, algo, split)
    return hook


res = {}
for arm in arms:
    mods = wct_oracle.Modules(mode, w)
    mods.conv_hook = hook_for(arm)
    t0 = time.time()
    out = wct_oracle.stylize(mods, c, s, 1.0)
    r = compare_to_fixture(out, g)
    res[arm] = out
    print("%s %-10s vs reference: max %.3e  p99.99 %.3e  frac>1e-3 %.2e  down16 %.2e   (%d emulated layer calls, %.0f s)"
          % (frame, arm, r["max"], r["lattice_p9999"], r["lattice_frac_gt_gate"], r["down16_max"], count.get(arm, 0), time.time() - t0), flush=True)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
if "oracle" in res:
    for arm in arms:
        if arm != "oracle":
            print("%s %-10s vs oracle arm: %.3e" % (frame, arm, rel(res[arm], res["oracle"])))
