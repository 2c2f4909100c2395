"""Bitwise A/B of one build under two environments: python tools/experiments/env_equal.py "VAR=1 OTHER=2"  (runs itself twice)."""
import os, subprocess, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--run":
    sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
    import numpy as np, torch
    from wct_hip import WCT, model_zoo
    w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
    g = torch.Generator(device="cuda").manual_seed(3)
    outs = []
    for (h, wd, hs, ws, a) in ((333, 517, 200, 260, 1.0), (1083, 1925, 1024, 1024, 0.7), (2160, 3840, 512, 512, 1.0)):
        c, s = torch.rand((3, h, wd), device="cuda", generator=g), torch.rand((3, hs, ws), device="cuda", generator=g)
        outs.append(wct.stylize(c, s, alpha=a).cpu().numpy())
    np.savez(sys.argv[2], *outs)
else:
    import numpy as np
    envb = dict(kv.split("=") for kv in sys.argv[1].split())
    for tag, extra in (("a", {}), ("b", envb)):
        subprocess.check_call([sys.executable, __file__, "--run", "/tmp/env_%s.npz" % tag], env=dict(os.environ, WCT_DEBUG="1", **extra))
    a, b = np.load("/tmp/env_a.npz"), np.load("/tmp/env_b.npz")
    for k in a.files:
        print(k, a[k].shape, "bitwise equal" if np.array_equal(a[k], b[k]) else "DIFFER max %.3e" % np.abs(a[k] - b[k]).max())
