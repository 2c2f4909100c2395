"""Decoders with the upsample layers on the low-resolution grid vs the nine-tap form (WCT_TAIL_UP=0 WCT_SP_UP=0): where do they differ?  python tools/experiments/tail_up_debug.py"""
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
    for (h, wd) in ((24, 40), (64, 96), (540, 960)):
        f = torch.rand((1, 32, h, wd), device="cuda", generator=g)
        outs.append(wct.d2(f).cpu().numpy())
    for (h, wd) in ((7, 11), (33, 50), (135, 240)):
        f = torch.rand((1, 128, h, wd), device="cuda", generator=g)
        outs.append(wct.d5(f).cpu().numpy())
    np.savez(sys.argv[2], *outs)
else:
    import numpy as np
    for tag, extra in (("a", {}), ("b", {"WCT_TAIL_UP": "0", "WCT_SP_UP": "0"})):
        subprocess.check_call([sys.executable, __file__, "--run", "/tmp/tu_%s.npz" % tag], env=dict(os.environ, WCT_DEBUG="1", **extra))
    a, b = np.load("/tmp/tu_a.npz"), np.load("/tmp/tu_b.npz")
    for k in a.files:
        d = np.abs(a[k] - b[k])[0].max(axis=0)
        print(k, a[k].shape, "max diff %.3e (ref max %.3e)" % (d.max(), np.abs(b[k]).max()))
        bad = np.argwhere(d > 1e-4 * np.abs(b[k]).max())
        print("  bad pixels:", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:40], "cols", sorted(set(bad[:, 1].tolist()))[:60])
