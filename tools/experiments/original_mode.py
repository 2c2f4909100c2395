"""BASELINE config 3: --mode original (un-pruned VGG-19 graph, generated weights), 1920x1080 content + style."""
import os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "collaborative-distillation_amd"))
import torch
from wct_hip import WCT, model_zoo
w = model_zoo.synth_weights("original", 2099)
wct = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=w)
g = torch.Generator(device="cuda").manual_seed(3)
c = torch.rand((3, 1080, 1920), device="cuda", generator=g); s = torch.rand((3, 1080, 1920), device="cuda", generator=g)
for i in range(2):
    out = wct.stylize(c, s); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3):
    out = wct.stylize(c, s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("original mode 1920x1080: %.1f ms/step  %.1f MP/s  out %s finite=%s range [%.3f, %.3f]" % (dt * 1e3, 1920 * 1080 / 1e6 / dt, tuple(out.shape), bool(torch.isfinite(out).all()), out.min().item(), out.max().item()))
wct.set_overlap(False); wct.profile_reset(); wct.profile(True); wct.stylize(c, s); torch.cuda.synchronize(); wct.profile(False)
for e in sorted(wct.profile_read(), key=lambda e: -e["ms"])[:30]:
    print("  %-34s %8.3f ms  %3d launches  %s" % (e["name"], e["ms"], e["launches"], ("%.0f TF" % (e["flops"] / e["ms"] / 1e9)) if e["flops"] else ""))
