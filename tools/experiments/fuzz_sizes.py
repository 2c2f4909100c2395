"""Random image sizes, both modes, every level in isolation on the checker's own level inputs + the cascade: a sweep for
shape-dependent bugs (tile/group distribution, partial tiles, tiny feature maps, singular covariances)."""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np, torch
from oracle import wct_oracle
from wct_hip import WCT, model_zoo
wct_oracle.set_num_threads(32)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
worst = 0.0
for mode, ncase, hi in (("16x", 14, 900), ("original", 5, 300)):
    w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz")) if mode == "16x" else model_zoo.synth_weights("original", 7)
    mods = wct_oracle.Modules(mode, w)
    wct = WCT(types.SimpleNamespace(mode=mode, alpha=1.0), weights=w)
    for case in range(ncase):
        H, W, Hs, Ws = (int(v) for v in rng.integers(32, hi, 4))
        alpha = float(rng.choice([1.0, 0.6]))
        c = rng.random((3, H, W), dtype=np.float32); s = rng.random((3, Hs, Ws), dtype=np.float32)
        trace = []
        ref = wct_oracle.stylize(mods, c, s, alpha, trace=trace)
        iso, img = [], c
        for t in trace:
            g = wct.style_transfer_level(t["level"], torch.from_numpy(img).cuda(), torch.from_numpy(s).cuda(), alpha=alpha).cpu().numpy()[0]
            assert g.shape == t["out"].shape, (g.shape, t["out"].shape)
            iso.append(float(np.abs(g - t["out"]).max() / np.abs(t["out"]).max()))
            img = t["out"]
        got = wct.stylize(torch.from_numpy(c).cuda(), torch.from_numpy(s).cuda(), alpha=alpha).cpu().numpy()[0]
        e2e = float(np.abs(got - ref).max() / np.abs(ref).max())
        worst = max(worst, max(iso))
        flag = "  <-- CHECK" if max(iso) > 2e-4 or not np.isfinite(got).all() else ""
        print("%-8s %4dx%-4d style %4dx%-4d alpha %.1f: level-isolated %s  e2e %.1e%s" % (mode, H, W, Hs, Ws, alpha, " ".join("%.0e" % v for v in iso), e2e, flag), flush=True)
print("worst level-isolated error %.2e" % worst)
