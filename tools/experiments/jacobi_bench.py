"""Time wct_solve (two eigen-decompositions + assembly) on synthetic covariances: prints sweeps and ms."""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "collaborative-distillation_amd"))
import numpy as np, torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
rng = np.random.default_rng(0)
for C, dead in ((128, 0), (128, 29), (128, 77), (64, 5), (32, 4), (24, 0)):
    n = 20000
    base = rng.standard_normal((n, 16)).astype(np.float32)
    X = np.maximum(base @ rng.standard_normal((16, C)).astype(np.float32) + 0.7 * rng.standard_normal((n, C)).astype(np.float32), 0)
    X[:, :dead] = 0
    f = torch.from_numpy(X.reshape(100, 200, C)).cuda()[None]
    nc, s, ss = wct.moments(f)
    M, b, info = wct.solve(nc, s, ss, nc, s, ss, 1.0, want_info=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        wct.solve(nc, s, ss, nc, s, ss, 1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nl = C - dead
    print("C=%3d live=%3d sweeps=%s  solve(2 eig + assemble) %.3f ms  -> per eig ~%.3f ms, per round ~%.2f us" % (
        C, nl, info, ms, ms / 2, ms / 2 / max(1, (nl - 1) * info[0]) * 1e3))
    # identity check: M should be ~I when content == style
    print("      |M - P|max = %.2e" % (M - torch.diag((torch.arange(C, device='cuda') >= dead).double())).abs().max().item())
