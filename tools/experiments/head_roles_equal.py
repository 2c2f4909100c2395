import os, sys, types, hashlib
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
g = torch.Generator(device="cuda").manual_seed(3)
out = []
for (H, W) in ((1100, 1950), (2160, 3840), (2048, 2048), (1030, 4100)):
    c = torch.rand((1, 3, H, W), device="cuda", generator=g)
    for L in (5, 2):
        y = wct.encode(L, c)
        out.append("%dx%d L%d %s %s" % (H, W, L, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16], bool(torch.isfinite(y).all())))
print("\n".join(out))
