"""BASELINE config 4 on ONE GPU: 10240x4096 content, 2048x2048 style (the multi-GPU config, untiled): finite output,
time per frame, peak memory."""
import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT
w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0))
g = torch.Generator(device="cuda").manual_seed(5)
c = torch.rand((1, 3, 4096, 10240), device="cuda", generator=g)
s = torch.rand((1, 3, 2048, 2048), device="cuda", generator=g)
out = w.stylize(c, s); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = w.stylize(c, s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
free, total = torch.cuda.mem_get_info()
print("cfg4 1 GPU: %.1f ms/frame = %.0f MP/s, finite=%s, min %.3f max %.3f, device memory in use %.1f GB" % (
    dt * 1e3, 41.94 / dt, bool(torch.isfinite(out).all()), float(out.min()), float(out.max()), (total - free) / 2**30))
