"""GPU box (round 4, VERDICT r3 task 1a): is the K = 27 split-f16 first convolution measurably worse than exact-fp32 MFMA?
A/B of the first conv's arithmetic on the reference-made fixtures:
  config 3 (G14, He-uniform generated weights; G15, conditioned weights): debug key in3wide = 2 (exact-fp32 MFMA, in3_wide_f32_kernel: the default) / 1 (f16x3, in3_wide_kernel) / 0 (exact-fp32
      MFMA, generic kernel): end to end against the reference's pixels, and level-isolated against the oracle's fp64 arm (each level on the truth's
      own level input);
  config 2 (G13 noise): default (fused head: conv11 3->16 in K-concatenated f16x3) / fuse = 0 (conv11 on the exact-fp32 MFMA kernel) /
      l1fuse = 0 (level 1's 3->24 the same way).
usage: python tools/experiments/first_conv_ab.py [--no-truth]  -> gpurun_out/first_conv_ab.txt"""
import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np, torch
from oracle import wct_oracle
from wct_hip import WCT, model_zoo
from tests.conftest import GOLD, PKG, load_golden, rel_err
from tests.fixture_compare import cfg2_frames, cfg3_frames, compare_to_fixture

os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
out = open(os.path.join(REPO, "gpurun_out", "first_conv_ab.txt"), "w")


def say(*a):
    line = " ".join(str(v) for v in a)
    print(line, flush=True); out.write(line + "\n"); out.flush()


cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
wct_oracle.set_num_threads(min(os.cpu_count() or 1, 32))
truth = "--no-truth" not in sys.argv
c, s = cfg3_frames()
for tag, w, gname in (("G14 he-uniform", model_zoo.synth_weights("original", 3), "g14_cfg3_original.npz"),
                      ("G15 conditioned", model_zoo.synth_weights_conditioned("original", 15), "g15_cfg3_conditioned_noise.npz")):
    g = load_golden(gname)
    t64 = []
    if truth:
        t0 = time.time()
        wct_oracle.stylize(wct_oracle.Modules("original", w, precision="fp64"), c, s, 1.0, trace=t64)
        say("%s: fp64 truth trace %.0f s" % (tag, time.time() - t0))
    eng = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=w)
    for key in (2, 1, 0):
        eng.debug_set("in3wide", key)
        got = eng.stylize(cu(c), cu(s)).cpu().numpy()[0]
        r = compare_to_fixture(got, g)
        iso = []
        img = c
        for t in t64:
            y = eng.style_transfer_level(t["level"], cu(np.asarray(img, np.float32)), cu(s)).cpu().numpy()[0]
            iso.append(rel_err(y, t["out"]))
            img = t["out"]
        say("%s in3wide=%d (%s): hip_vs_reference %.4e  p99.99 %.3e  frac>1e-3 %.2e  down16 %.2e | level-isolated vs fp64 truth L5..L1: %s"
            % (tag, key, {2: "exact fp32 MFMA, wide kernel (default)", 1: "f16x3 K=27", 0: "exact fp32 MFMA, generic kernel"}[key], r["max"], r["lattice_p9999"], r["lattice_frac_gt_gate"], r["down16_max"],
               " ".join("%.2e" % v for v in iso)))
    del eng
w16 = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
g13 = load_golden("g13_cfg2_noise.npz")
c2, s2 = cfg2_frames("noise")
for name, keys in (("default (fused head, f16x3 K=27 first conv)", {}), ("fuse=0 (3->16 first conv on exact-fp32 MFMA)", {"fuse": 0}),
                   ("l1fuse=0 (level 1's 3->24 on exact-fp32 MFMA)", {"l1fuse": 0}), ("fuse=0 l1fuse=0", {"fuse": 0, "l1fuse": 0})):
    eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w16)
    for k, v in keys.items():
        eng.debug_set(k, v)
    got = eng.stylize(cu(c2), cu(s2)).cpu().numpy()[0]
    r = compare_to_fixture(got, g13)
    say("cfg2 G13 noise, %s: hip_vs_reference %.4e  p99.99 %.3e  down16 %.2e" % (name, r["max"], r["lattice_p9999"], r["down16_max"]))
    del eng
