"""Throughput with several content frames in flight on one GPU (one context + one torch stream per frame slot): the
matrix-function stretch of a level leaves the chip idle on the critical path of ONE frame; another frame's kernels can fill it."""
import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
H, W = 2160, 3840
style = torch.rand((3, 2048, 2048), device="cuda")
for nslots in (1, 2, 3):
    ctxs = [WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w) for _ in range(nslots)]
    streams = [torch.cuda.Stream() for _ in range(nslots)]
    frames = [torch.rand((3, H, W), device="cuda") for _ in range(nslots)]
    outs = [torch.empty((3, H, W), device="cuda") for _ in range(nslots)]
    for cached in (False, True):
        if cached:
            for c in ctxs: c.style_prepare(style)
        torch.cuda.synchronize()
        def run(n):
            for i in range(n):
                k = i % nslots
                with torch.cuda.stream(streams[k]):
                    if cached: ctxs[k].stylize_prepared(frames[k])
                    else: ctxs[k].stylize(frames[k], style, out=outs[k])
        run(2 * nslots); torch.cuda.synchronize()
        n = 12 * nslots
        t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("frames in flight %d, style %s: %.2f ms per frame, %.0f MP/s" % (nslots, "cached" if cached else "per frame", dt / n * 1e3, H * W / 1e6 * n / dt))
    del ctxs
