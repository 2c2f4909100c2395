"""Solver accuracy against numpy eigh on covariances with a prescribed log-uniform spectrum 1 .. lmin."""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np, torch
from oracle import wct_oracle as oracle
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
for C in (128, 256, 512):
    for lmin in (1e-6, 1e-9, 1e-11, 1e-13, 1e-15):
        rng = np.random.default_rng(C)
        def spd(lo, scale):
            Q, _ = np.linalg.qr(rng.standard_normal((C, C)))
            lam = scale * np.exp(np.linspace(0.0, np.log(lo), C))
            A = (Q * lam) @ Q.T
            return (A + A.T) / 2
        cov_c, cov_s = spd(lmin, 40.0), spd(1e-6, 3.0)
        mu_c, mu_s = rng.random(C), rng.random(C)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64)).cuda()
        raw = lambda n, mu, cov: (n, dev(n * mu), dev((n - 1) * cov + n * np.outer(mu, mu)))
        M, b, info = wct.solve(*raw(50000.0, mu_c, cov_c), *raw(20000.0, mu_s, cov_s), alpha=1.0, want_info=True)
        M = M.cpu().numpy()
        Mr, br = oracle.affine_from_moments(mu_c, cov_c, mu_s, cov_s, 1.0)
        Mk, _ = oracle.affine_from_moments(mu_c, cov_c, mu_s, cov_s, 1.0, rel_thresh=1e-14, abs_floor=1e-16)
        print("C %3d lmin %.0e info %s  |M| %.1e  err vs policy %.1e  vs keep-all %.1e  b err %.1e" % (C, lmin, info, np.abs(Mr).max(), rel(M, Mr), rel(M, Mk), rel(b.cpu().numpy(), br)))
