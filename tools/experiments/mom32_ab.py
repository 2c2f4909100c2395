"""GPU box (round 4, VERDICT r3 task 7): end-to-end parity of the fp32-block-product moments (debug key mom32: 0 = fp64 products, 1 =
fp32 blocks on maps of >= 65 536 pixels (default), 2 = fp32 blocks at every size) against the reference's own pixels:
G13 config-2 noise / smooth, G11 UHD pair (16x), G15 noise + G14 (--mode original).   -> gpurun_out/mom32_ab.txt"""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np, torch
from wct_hip import WCT, model_zoo
from tests.conftest import GOLD, PKG, load_golden
from tests.fixture_compare import cfg2_frames, cfg3_frames, compare_to_fixture

os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
out = open(os.path.join(REPO, "gpurun_out", "mom32_ab.txt"), "w")


def say(*a):
    line = " ".join(str(v) for v in a)
    print(line, flush=True); out.write(line + "\n"); out.flush()


cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
w16 = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
jobs = [("cfg2 noise (G13)", "16x", w16, cfg2_frames("noise"), "g13_cfg2_noise.npz"),
        ("cfg2 smooth (G13)", "16x", w16, cfg2_frames("smooth"), "g13_cfg2_smooth.npz"),
        ("cfg3 conditioned (G15 noise)", "original", model_zoo.synth_weights_conditioned("original", 15), cfg3_frames(), "g15_cfg3_conditioned_noise.npz"),
        ("cfg3 he-uniform (G14)", "original", model_zoo.synth_weights("original", 3), cfg3_frames(), "g14_cfg3_original.npz")]
for tag, mode, w, (c, s), gname in jobs:
    g = load_golden(gname)
    eng = WCT(types.SimpleNamespace(mode=mode, alpha=1.0), weights=w)
    base = None
    for m in (0, 1, 2):
        eng.debug_set("mom32", m)
        got = eng.stylize(cu(c), cu(s)).cpu().numpy()[0]
        r = compare_to_fixture(got, g)
        if base is None:
            base = got
        say("%s mom32=%d: hip_vs_reference %.4e  p99.99 %.3e  frac>1e-3 %.2e  down16 %.2e | vs mom32=0: %.3e"
            % (tag, m, r["max"], r["lattice_p9999"], r["lattice_frac_gt_gate"], r["down16_max"], float(np.abs(got - base).max() / np.abs(base).max())))
    del eng
