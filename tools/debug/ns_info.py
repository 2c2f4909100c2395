import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "collaborative-distillation_amd"))
import numpy as np, torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
g = torch.Generator(device="cuda").manual_seed(1); c = torch.rand((3, 2160, 3840), device="cuda", generator=g)
g2 = torch.Generator(device="cuda").manual_seed(2); s = torch.rand((3, 2048, 2048), device="cuda", generator=g2)
img = c[None]
for L in (5, 4, 3, 2, 1):
    cF = wct.encode(L, img, layout="nhwc"); sF = wct.encode(L, s, layout="nhwc")
    nc, sc, ssc = wct.moments(cF); ns, ss, sss = wct.moments(sF)
    M, b, info = wct.solve(nc, sc, ssc, ns, ss, sss, 1.0, want_info=True)
    def spec(n, s1, s2):
        mu = s1 / n; cov = ((s2 - n * torch.outer(mu, mu)) / (n - 1)).cpu().numpy()
        ex2 = (np.diag(cov) + mu.cpu().numpy() ** 2).max(); live = np.diag(cov) > 1e-13 * ex2
        lam = np.linalg.eigvalsh(cov[np.ix_(live, live)]); return live.sum(), lam[-1] / max(lam[0], 1e-300), lam[0] / lam[-1]
    print("L%d info(content,style)=%s content(live,cond)=%s style(live,cond)=%s" % (L, info, spec(nc, sc, ssc)[:2], spec(ns, ss, sss)[:2]))
    img = wct.decode_affine(L, cF, M, b)
