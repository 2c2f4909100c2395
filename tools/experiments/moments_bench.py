import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "collaborative-distillation_amd"))
import torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
for (h, wd, C) in ((2160, 3840, 24), (1080, 1920, 32), (540, 960, 64), (270, 480, 128), (135, 240, 128), (2048, 2048, 24)):
    f = torch.rand((1, h, wd, C), device="cuda")
    wct.moments(f); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): wct.moments(f)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("h=%d w=%d C=%d: %.1f us  %.0f GB/s  (%.1f f64-TF)" % (h, wd, C, ms * 1e3, h * wd * C * 4 / ms / 1e6, 2.0 * C * C * h * wd / ms / 1e9 / 2))
