"""CPU only: how far is the reference's own arithmetic (fp32 conv stacks, fp64 transform = oracle precision "fp32") from the
exact result (every activation and accumulation in fp64 = precision "fp64") after the 5-level cascade?  This is the yardstick
for the HIP path's end-to-end figure: two valid fp32 implementations of the reference differ by about this much."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np
from oracle import wct_oracle
from wct_hip import model_zoo

def smooth(rng, shape, it=3):
    x = rng.random(shape, dtype=np.float32)
    for _ in range(it):
        x = (x + np.roll(x, 1, 1) + np.roll(x, 1, 2) + np.roll(x, -1, 1) + np.roll(x, -1, 2)) / 5
    return np.ascontiguousarray((x - x.min()) / (x.max() - x.min()))

mode = sys.argv[1] if len(sys.argv) > 1 else "16x"
H, W, Hs, Ws = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (1080, 1920, 1024, 1024)
kind = sys.argv[6] if len(sys.argv) > 6 else "noise"
seed = int(sys.argv[7]) if len(sys.argv) > 7 else 7
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz")) if mode == "16x" else model_zoo.synth_weights("original", seed)
wct_oracle.set_num_threads(min(os.cpu_count(), 32))
rng = np.random.default_rng(0)
c = rng.random((3, H, W), dtype=np.float32) if kind == "noise" else smooth(rng, (3, H, W))
s = rng.random((3, Hs, Ws), dtype=np.float32)
m32, m64 = wct_oracle.Modules(mode, w), wct_oracle.Modules(mode, w, precision="fp64")
t32, t64 = [], []
t0 = time.time(); r32 = wct_oracle.stylize(m32, c, s, 1.0, trace=t32); t1 = time.time()
r64 = wct_oracle.stylize(m64, c, s, 1.0, trace=t64); t2 = time.time()
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
print("%s %dx%d %s: fp32 %.1f s, fp64 %.1f s" % (mode, W, H, kind, t1 - t0, t2 - t1))
print("cumulative |fp32 - truth| after L5..L1:", " ".join("%.2e" % rel(a["out"], b["out"]) for a, b in zip(t32, t64)))
# level-isolated: fp32 level on truth's (rounded) input vs truth's output
iso = []
img = c
for b in t64:
    o = wct_oracle.style_transfer(m32, b["level"], np.asarray(img, np.float32), s, 1.0)
    iso.append(rel(o, b["out"]))
    img = b["out"]
print("level-isolated |fp32 - truth| L5..L1:  ", " ".join("%.2e" % v for v in iso))
print("end to end: %.3e" % rel(r32, r64))
