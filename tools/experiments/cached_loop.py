import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
c = torch.rand((3, 2160, 3840), device="cuda"); s = torch.rand((3, 2048, 2048), device="cuda")
wct.style_prepare(s)
for _ in range(4):
    wct.stylize_prepared(c)
torch.cuda.synchronize()
