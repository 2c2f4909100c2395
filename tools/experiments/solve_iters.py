"""Iteration counts and times of the matrix-function solves on the benchmark's own covariances (4K noise content, 2K noise style): per level the
content (inverse square root) and style (square root) solve of wct_solve: Newton-Schulz iterations (info) and HIP-event time of the pair."""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch  # noqa: E402
from tests.fixture_compare import noise_frame, smooth_frame  # noqa: E402
import numpy as np  # noqa: E402
from wct_hip import WCT, model_zoo  # noqa: E402

w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
for kind in ("noise", "smooth"):
    c = torch.from_numpy(noise_frame(1, 2160, 3840) if kind == "noise" else smooth_frame(np.random.default_rng(101), (3, 2160, 3840))).cuda()
    s = torch.from_numpy(noise_frame(2, 2048, 2048)).cuda()
    img = c[None]
    for L in (5, 4, 3, 2, 1):
        fc = eng.encode(L, img, layout="nhwc")
        fs = eng.encode(L, s[None], layout="nhwc")
        nc, sc, qc = eng.moments(fc)
        ns, ss, qs = eng.moments(fs)
        M, b, info = eng.solve(nc, sc, qc, ns, ss, qs, want_info=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.solve(nc, sc, qc, ns, ss, qs)
        e1.record()
        torch.cuda.synchronize()
        cov = (qc - torch.outer(sc, sc) / nc) / (nc - 1)
        ev = torch.linalg.eigvalsh(cov.cpu())
        live = ev[ev > ev.max() * 1e-12]
        print("%s level %d C=%3d: iterations content %d / style %d; solve pair + assemble %.1f us; content cov cond (live) %.2e, dead %d" % (
            kind, L, int(sc.numel()), info[0], info[1], e0.elapsed_time(e1) * 100, float(live.max() / live.min()), int((ev <= ev.max() * 1e-12).sum())), flush=True)
        img = eng.decode_affine(L, fc, M, b)
