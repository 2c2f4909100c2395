"""Bitwise A/B of two builds of libwct_hip: python tools/experiments/ab_equal.py <other.so>  (runs itself twice, compares)."""
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
    for (h, wd, hs, ws, a) in ((333, 517, 200, 260, 1.0), (1080, 1920, 1024, 1024, 0.7)):
        c, s = torch.rand((3, h, wd), device="cuda", generator=g), torch.rand((3, hs, ws), device="cuda", generator=g)
        outs.append(wct.stylize(c, s, alpha=a).cpu().numpy())
    wo = model_zoo.synth_weights("original", 7)
    wct2 = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=wo)
    c, s = torch.rand((3, 160, 208), device="cuda", generator=g), torch.rand((3, 128, 96), device="cuda", generator=g)
    outs.append(wct2.stylize(c, s).cpu().numpy())
    np.savez(sys.argv[2], *outs)
else:
    import numpy as np
    other = sys.argv[1]
    for tag, lib in (("a", ""), ("b", other)):
        subprocess.check_call([sys.executable, __file__, "--run", "/tmp/ab_%s.npz" % tag], env=dict(os.environ, WCT_LIB_PATH=lib))
    a, b = np.load("/tmp/ab_a.npz"), np.load("/tmp/ab_b.npz")
    for k in a.files:
        print(k, a[k].shape, "bitwise equal" if np.array_equal(a[k], b[k]) else "DIFFER max %.3e" % np.abs(a[k] - b[k]).max())
