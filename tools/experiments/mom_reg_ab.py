"""Round 5: moments_reg_kernel (operands straight from the loads' registers) against the LDS kernel's fp32-block form, same process layout as
ab_equal.py: runs itself twice (WCT_DEBUG=1 WCT_MOM_REG=0 / 1), compares raw sums and times the calls.  python tools/experiments/mom_reg_ab.py"""
import os, subprocess, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODE = "blk" if "blk" in sys.argv[1:] else "reg"
ENVVAR = {"reg": "WCT_MOM_REG", "blk": "WCT_MOM_BLK"}[MODE]
CASES_BLK = [(512, 134, 240, 0, None), (512, 67, 120, 0, None), (512, 134, 240, 16, 231), (512, 128, 128, 0, None), (512, 100, 77, 3, 70),
             (128, 536, 960, 0, None), (256, 268, 480, 0, None), (128, 270, 480, 0, None), (128, 256, 256, 0, None), (256, 268, 480, 9, 400), (128, 536, 960, 100, 333), (512, 256, 300, 0, None)]
CASES = CASES_BLK if MODE == "blk" else [(32, 1080, 1920, 0, None), (32, 1024, 1024, 0, None), (64, 540, 960, 0, None), (64, 512, 512, 0, None), (32, 1080, 1920, 640, 1237), (64, 540, 960, 3, 701),
         (32, 300, 333, 0, None), (64, 270, 487, 5, 480)]
CASES = CASES_BLK if MODE == "blk" else CASES
if len(sys.argv) > 2 and sys.argv[1] == "--run":
    sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
    import numpy as np, torch
    from wct_hip import WCT, model_zoo
    w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
    g = torch.Generator(device="cuda").manual_seed(3)
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i, (C, h, wd, x0, x1) in enumerate(CASES):
        f = torch.rand((1, h, wd, C), device="cuda", generator=g) * 3 - 0.5
        args = (f,) if x1 is None else (f, x0, x1)
        n, s1, s2 = wct.moments(*args)
        for _ in range(3):
            wct.moments(*args)
        e0.record()
        for _ in range(10):
            wct.moments(*args)
        e1.record(); torch.cuda.synchronize()
        out["n%d" % i], out["s%d" % i], out["q%d" % i], out["t%d" % i] = np.float64(n), s1.cpu().numpy(), s2.cpu().numpy(), np.float64(e0.elapsed_time(e1) / 10 * 1e3)
        f64 = f[0].double().reshape(-1, C) if x1 is None else f[0, :, x0:x1].double().reshape(-1, C)
        out["r%d" % i] = (f64.t() @ f64).cpu().numpy()
    np.savez(sys.argv[2], **out)
else:
    import numpy as np
    for tag, v in (("lds", "0"), ("reg", "1")):
        subprocess.check_call([sys.executable, __file__, "--run", "/tmp/momab_%s.npz" % tag] + sys.argv[1:], env=dict(os.environ, **{"WCT_DEBUG": "1", ENVVAR: v}))
    a, b = np.load("/tmp/momab_lds.npz"), np.load("/tmp/momab_reg.npz")
    rel = lambda x, y: float(np.abs(x - y).max() / np.abs(y).max())
    for i, (C, h, wd, x0, x1) in enumerate(CASES):
        win = "" if x1 is None else " cols [%d,%d)" % (x0, x1)
        print("C=%-3d %4dx%-4d%-18s LDS kernel %7.1f us  register kernel %7.1f us (%.2fx)   sums %s  sumsq %s (rel %.1e)   vs torch fp64: lds %.1e reg %.1e   %.0f -> %.0f GB/s" % (
            C, h, wd, win, a["t%d" % i], b["t%d" % i], a["t%d" % i] / b["t%d" % i],
            "bitwise" if np.array_equal(a["s%d" % i], b["s%d" % i]) else "rel %.1e" % rel(b["s%d" % i], a["s%d" % i]),
            "bitwise" if np.array_equal(a["q%d" % i], b["q%d" % i]) else "differ", rel(b["q%d" % i], a["q%d" % i]),
            rel(a["q%d" % i], a["r%d" % i]), rel(b["q%d" % i], b["r%d" % i]),
            float(a["n%d" % i]) * C * 4 / a["t%d" % i] / 1e3, float(b["n%d" % i]) * C * 4 / b["t%d" % i] / 1e3))
