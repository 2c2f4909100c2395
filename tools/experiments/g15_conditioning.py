"""CPU only (round 4, VERDICT r3 task 1b): find an `--mode original` frame / generated weight set on which the reference's OWN
fp32 arithmetic is not chaotic, so that the un-pruned graph can be gated end to end at the literal 1e-3 (fixture G15).

Two valid fp32 implementations of the reference's op sequence -- torch CPU convolutions (what the reference runs) and the oracle's
C loops -- are run through the 5-level cascade; their distance is the proxy for `oracle_vs_reference`.  Per level the condition of
the content / style covariances is printed (whitening multiplies any fp32-level difference by sqrt(lambda_s_max / lambda_c_min)).

usage: python tools/experiments/g15_conditioning.py <weights: he|wc> <input: noise|natural|smooth> H W [seed]
"""
import os, sys, time
REPO = "/root/repo"; sys.path[:0] = [REPO, REPO + "/collaborative-distillation_amd"]
import numpy as np, torch, torch.nn.functional as F
from oracle import wct_oracle
from wct_hip import model_zoo
from tests.fixture_compare import smooth_frame

torch.set_num_threads(8)
wkind, ikind, H, W = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 3
w = model_zoo.synth_weights("original", seed) if wkind == "he" else model_zoo.synth_weights_conditioned("original", seed)


def natural(path, h, wd):
    from PIL import Image
    im = Image.open(path).convert("RGB").resize((wd, h), Image.BILINEAR)
    return np.ascontiguousarray(np.asarray(im).transpose(2, 0, 1).astype(np.float32) / np.float32(255))


if ikind == "noise":
    c, s = np.random.default_rng(3).random((3, H, W), dtype=np.float32), np.random.default_rng(4).random((3, H, W), dtype=np.float32)
elif ikind == "smooth":
    c, s = smooth_frame(np.random.default_rng(3), (3, H, W)), smooth_frame(np.random.default_rng(4), (3, H, W))
else:
    c = natural(REPO + "/tests/golden/g11_uhd_content_3840x2160.jpg", H, W)
    s = natural(REPO + "/tests/golden/g11_style_2048x2048.jpg", H, W)


class TorchMods:
    precision = "fp32"

    def conv(self, x, wt, b):
        return F.relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), torch.from_numpy(wt), torch.from_numpy(b)))

    def encode(self, level, img):
        key = "e%d" % level
        y = torch.from_numpy(np.asarray(img, np.float32))[None]
        y = F.conv2d(y, torch.from_numpy(w[key + ".conv0.weight"]), torch.from_numpy(w[key + ".conv0.bias"]))
        for l in model_zoo.encoder_layers("original", level):
            y = self.conv(y, w["%s.%s.weight" % (key, l.name)], w["%s.%s.bias" % (key, l.name)])
            if l.pool_after:
                y = F.max_pool2d(y, 2, 2)
        return y[0].numpy()

    def decode(self, level, feat):
        key = "d%d" % level
        y = torch.from_numpy(np.asarray(feat, np.float32))[None]
        for l in model_zoo.decoder_layers("original", level):
            y = self.conv(y, w["%s.%s.weight" % (key, l.name)], w["%s.%s.bias" % (key, l.name)])
            if l.up_after:
                y = F.interpolate(y, scale_factor=2, mode="nearest")
        return y[0].numpy()


rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
with torch.no_grad():
    t0 = time.time()
    tt = []
    r_t = wct_oracle.stylize(TorchMods(), c, s, 1.0, trace=tt)
    t1 = time.time()
    wct_oracle.set_num_threads(8)
    to = []
    r_o = wct_oracle.stylize(wct_oracle.Modules("original", w), c, s, 1.0, trace=to)
    t2 = time.time()
for a, b in zip(tt, to):
    _, _, cc = wct_oracle.moments(a["cF"]); _, _, cs = wct_oracle.moments(a["sF"])
    lc, ls = np.linalg.eigvalsh(cc), np.linalg.eigvalsh(cs)
    print("L%d  C=%d hw=%d  content lam max %.3e min %.3e (cond %.1e)  style max %.3e min %.3e  amp sqrt(ls_max/lc_min) %.1e | out torch-vs-oracle %.2e  max %.3f"
          % (a["level"], cc.shape[0], a["cF"][0].size, lc.max(), lc.min(), lc.max() / max(lc.min(), 1e-300), ls.max(), ls.min(),
             np.sqrt(ls.max() / max(lc.min(), 1e-300)), rel(b["out"], a["out"]), np.abs(a["out"]).max()), flush=True)
print("%s %s %dx%d seed %d: torch32 vs oracle32 END TO END %.3e   (torch %.0f s, oracle %.0f s)  out mean %.4f max %.4f min %.4f"
      % (wkind, ikind, H, W, seed, rel(r_o, r_t), t1 - t0, t2 - t1, r_t.mean(), r_t.max(), r_t.min()))
