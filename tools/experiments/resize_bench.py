"""Device Resize timing (HIP events on the context's stream = torch's current stream): python tools/experiments/resize_bench.py"""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
g = torch.Generator(device="cuda").manual_seed(0)
for (h, wd, target) in ((2160, 3840, (1080, 1920)), (2160, 3840, (512, 910)), (4096, 10240, (2160, 5400)), (2048, 2048, (1024, 1024)), (1080, 1920, (2160, 3840))):
    x = torch.randint(0, 256, (h, wd, 3), device="cuda", dtype=torch.uint8, generator=g)
    for to_tensor in (False, True):
        for _ in range(3):
            wct.resize_u8(x, target, to_tensor=to_tensor)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            wct.resize_u8(x, target, to_tensor=to_tensor)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        by = h * wd * 3 + target[0] * target[1] * 3 * (4 if to_tensor else 1)
        print("%5dx%-5d -> %5dx%-5d %-9s %.3f ms  %.1f GB/s (in + out bytes)" % (h, wd, target[0], target[1], "planar" if to_tensor else "u8", ms, by / ms / 1e6))
