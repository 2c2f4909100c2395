"""--mode original (generated weights) against the CPU checker, level by level on the checker's own level inputs, at sizes
where the C = 512 covariances are full rank but ill-conditioned: does the Newton-Schulz result (larger budget) agree with
the reference arithmetic as well as the Jacobi fallback (WCT_NS_MAXIT=26 forces it for these matrices) does?"""
import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np, torch
from oracle import wct_oracle
from wct_hip import WCT, model_zoo
w = model_zoo.synth_weights("original", 2099)
wct_oracle.set_num_threads(32)
mods = wct_oracle.Modules("original", w)
wct = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=w)
for (H, W) in ((384, 384), (640, 768)):
    rng = np.random.default_rng(H)
    c = rng.random((3, H, W), dtype=np.float32); s = rng.random((3, 512, 512), dtype=np.float32)
    trace = []
    t0 = time.time(); ref = wct_oracle.stylize(mods, c, s, 1.0, trace=trace); t1 = time.time()
    iso = []
    img = c
    for t in trace:
        g = wct.style_transfer_level(t["level"], torch.from_numpy(img).cuda(), torch.from_numpy(s).cuda()).cpu().numpy()[0]
        iso.append(float(np.abs(g - t["out"]).max() / np.abs(t["out"]).max()))
        img = t["out"]
    got = wct.stylize(torch.from_numpy(c).cuda(), torch.from_numpy(s).cuda()).cpu().numpy()[0]
    print("%dx%d (oracle %.0f s): level-isolated L5..L1 %s; end-to-end %.2e" % (H, W, t1 - t0, " ".join("%.1e" % v for v in iso), np.abs(got - ref).max() / np.abs(ref).max()))
