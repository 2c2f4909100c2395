"""Why would the replica path (imported style statistics) differ from the single engine at 4K?  Bitwise comparisons on one GPU, one process."""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch  # noqa: E402
from tests.fixture_compare import noise_frame  # noqa: E402
from wct_hip import WCT, model_zoo  # noqa: E402

w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)   # noqa: E731
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
style = torch.from_numpy(noise_frame(2, 2048, 2048)).cuda()
c = torch.from_numpy(noise_frame(10, H, W)).cuda()
a = make()
a.style_prepare(style)
r1 = a.stylize_prepared(c).clone()
r2 = a.stylize_prepared(c).clone()
print("same engine twice:", bool(torch.equal(r1, r2)), float((r1 - r2).abs().max()))
b = make()
b.style_prepare(style)
r3 = b.stylize_prepared(c).clone()
print("second engine:", bool(torch.equal(r1, r3)), float((r1 - r3).abs().max()))
stats = {L: a.style_export(L).clone() for L in (5, 4, 3, 2, 1)}
d = make()
for L, v in stats.items():
    d.style_import(L, v)
r4 = d.stylize_prepared(c).clone()
print("imported statistics:", bool(torch.equal(r1, r4)), float((r1 - r4).abs().max()))
# level by level on the engine with imported statistics vs the preparing engine, same input image per level
img = c[None]
for L in (5, 4, 3, 2, 1):
    ha, wa, sa, qa = a.content_encode(L, img)
    hd, wd, sd, qd = d.content_encode(L, img)
    same_m = bool(torch.equal(sa, sd) and torch.equal(qa, qd))
    Ma, ba = a.content_solve(L, float(ha * wa), sa, qa)
    Md, bd = d.content_solve(L, float(hd * wd), sd, qd)
    same_s = bool(torch.equal(Ma, Md) and torch.equal(ba, bd))
    oa = a.content_decode(L, Ma, ba, int(img.shape[-2]), int(img.shape[-1]))
    od = d.content_decode(L, Md, bd, int(img.shape[-2]), int(img.shape[-1]))
    print("level %d: moments equal %s, (M, b) equal %s, decode equal %s (max %.3e)" % (L, same_m, same_s, bool(torch.equal(oa, od)), float((oa - od).abs().max())))
    img = oa
