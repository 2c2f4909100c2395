"""Newton-Schulz iteration counts / per-solve times on the benchmark's feature statistics."""
import os, sys, types, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT
w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0))
g = torch.Generator(device="cuda").manual_seed(0)
c = torch.rand((1, 3, 2160, 3840), device="cuda", generator=g)
s = torch.rand((1, 3, 2048, 2048), device="cuda", generator=g)
for L in (5, 4, 3, 2, 1):
    fc = w.encode(L, c, layout="nhwc"); fs = w.encode(L, s, layout="nhwc")
    nc, sc, qc = w.moments(fc); ns, ss, qs = w.moments(fs)
    M, b, info = w.solve(nc, sc, qc, ns, ss, qs, want_info=True)
    var = (torch.diagonal(qc) / nc - (sc / nc) ** 2)
    dead = int((var <= 1e-13 * (torch.diagonal(qc) / nc).max()).sum())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): w.solve(nc, sc, qc, ns, ss, qs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("L%d C=%d info=%s dead_content_channels=%d  solve(content+style+assemble) %.0f us" % (L, sc.numel(), info, dead, dt * 1e6))
