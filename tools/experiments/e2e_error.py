"""End-to-end error of the HIP path vs the CPU oracle on a mid-size sample, per conv mode: is the ~8e-4 of bench.py's
gpu_vs_oracle_rel_err the f16x3 arithmetic or the cascade's amplification of ANY fp32-level difference?"""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np, torch
from oracle import wct_oracle
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct_oracle.set_num_threads(32)
mods = wct_oracle.Modules("16x", w)
rng = np.random.default_rng(0)
c = rng.random((3, 1080, 1920), dtype=np.float32); s = rng.random((3, 1024, 1024), dtype=np.float32)
trace = []
ref = wct_oracle.stylize(mods, c, s, 1.0, trace=trace)
for mode in ("f16x3", "fp32"):
    os.environ["WCT_CONV_MODE"] = "1" if mode == "f16x3" else "0"
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
    got = wct.stylize(torch.from_numpy(c).cuda(), torch.from_numpy(s).cuda()).cpu().numpy()[0]
    e2e = np.abs(got - ref).max() / np.abs(ref).max()
    # level-isolated: feed the ORACLE's level input to the device
    iso = []
    img = c
    for t in trace:
        L = t["level"]
        g = wct.style_transfer_level(L, torch.from_numpy(img).cuda(), torch.from_numpy(s).cuda()).cpu().numpy()[0]
        iso.append(float(np.abs(g - t["out"]).max() / np.abs(t["out"]).max()))
        img = t["out"]
    print("%s: end-to-end %.2e; level-isolated L5..L1 %s" % (mode, e2e, " ".join("%.1e" % v for v in iso)))
