"""Iteration counts of the 4K bench frame's C = 128 solves and the cached-style frame time under WCT_NS_GUESS / WCT_NS_MAXIT."""
import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
g = torch.Generator(device="cuda").manual_seed(1)
c = torch.rand((3, 2160, 3840), device="cuda", generator=g)
g2 = torch.Generator(device="cuda").manual_seed(2)
s = torch.rand((3, 2048, 2048), device="cuda", generator=g2)
infos = []
for L in (5, 4):
    fc = wct.encode(L, c, layout="nhwc"); fs = wct.encode(L, s, layout="nhwc")
    nc, sc, ssc = wct.moments(fc); ns, ss, sss = wct.moments(fs)
    M, b, info = wct.solve(nc, sc, ssc, ns, ss, sss, alpha=1.0, want_info=True)
    infos.append(info)
wct.style_prepare(s)
for _ in range(3): wct.stylize_prepared(c)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): wct.stylize_prepared(c)
torch.cuda.synchronize()
print("GUESS=%s MAXIT=%s: iterations (content, style) L5 %s L4 %s; cached-style frame %.3f ms" % (os.environ.get("WCT_NS_GUESS", "-"), os.environ.get("WCT_NS_MAXIT", "-"), infos[0], infos[1], (time.perf_counter() - t0) / 20 * 1e3))
