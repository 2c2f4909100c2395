"""Per-family kernel times of the 4K step (overlap off, HIP events around every launch) -- for same-box A/Bs of environment-switched builds:
    WCT_DEBUG=1 WCT_L1DEC_TH=16 python tools/experiments/family_times.py l1_decode l1_moments
prints `family ms_per_step launches` for the families whose name contains one of the arguments (all if none)."""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tests.fixture_compare import noise_frame  # noqa: E402
from wct_hip import WCT, model_zoo  # noqa: E402

w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
c = torch.from_numpy(noise_frame(1, 2160, 3840)).cuda()
s = torch.from_numpy(noise_frame(2, 2048, 2048)).cuda()
out = torch.empty((3, 2160, 3840), device="cuda")
for _ in range(3):
    eng.stylize(c, s, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    eng.stylize(c, s, out=out)
e1.record()
torch.cuda.synchronize()
print("step_ms %.3f" % (e0.elapsed_time(e1) / 10))
eng.set_overlap(False)
eng.profile_reset()
eng.profile(True)
N = 5
for _ in range(N):
    eng.stylize(c, s, out=out)
torch.cuda.synchronize()
eng.profile(False)
rows = sorted(eng.profile_read(), key=lambda r: -r["ms"])
print("kernel_sum_ms %.3f" % (sum(r["ms"] for r in rows) / N))
for r in rows:
    if len(sys.argv) < 2 or any(k in r["name"] for k in sys.argv[1:]):
        print("%-40s %.4f %d" % (r["name"], r["ms"] / N, r["launches"] // N))
print("checksum %.9g" % float(out.double().sum()))
