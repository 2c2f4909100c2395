"""GPU diagnostic: per-level difference between the strip path and the untiled path (single process, no dist).
Rank r's all-reduce is emulated by summing both strips' moments computed in the same process."""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "collaborative-distillation_amd"))
import numpy as np, torch
from wct_hip import WCT, model_zoo
from wct_hip.sharded import CUM_HALO, LEVEL_HALO, ext_bounds, strip_bounds

w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
H, W, world = 272, 1525, 2
g = torch.Generator(device="cuda").manual_seed(11)
content = torch.rand((3, H, W), device="cuda", generator=g)
style = torch.rand((3, 300, 260), device="cuda", generator=g)
img_full = content[None]
W_cur = W
owns = strip_bounds(W, world)
imgs = [None] * world
los = [0] * world
for r in range(world):
    lo, hi = ext_bounds(owns[r], W, CUM_HALO[5]); imgs[r] = content[None, :, :, lo:hi].contiguous(); los[r] = lo
for L in (5, 4, 3, 2, 1):
    sh = L - 1
    sF = wct.encode(L, style, layout="nhwc"); ns, ss, sss = wct.moments(sF)
    cF = wct.encode(L, img_full, layout="nhwc"); nc, sc, ssc = wct.moments(cF)
    M, b, info = wct.solve(nc, sc, ssc, ns, ss, sss, 1.0, want_info=True)
    out_full = wct.decode_affine(L, cF, M, b)
    feats, sums, sqs = [], 0, 0
    for r in range(world):
        nlo, nhi = ext_bounds(owns[r], W_cur, CUM_HALO[L])
        imgs[r] = imgs[r][..., nlo - los[r]:nhi - los[r]].contiguous(); los[r] = nlo
        f = wct.encode(L, imgs[r], layout="nhwc")
        f0 = (owns[r][0] - nlo) >> sh
        f1 = f.shape[2] if owns[r][1] >= W_cur else (owns[r][1] - nlo) >> sh
        dfeat = (f[0, :, f0:f1] - cF[0, :, owns[r][0] >> sh:(owns[r][0] >> sh) + (f1 - f0)]).abs().max().item()
        _, s1, s2 = wct.moments(f, f0, f1)
        sums = sums + s1; sqs = sqs + s2; feats.append(f)
        print("L%d rank%d ext [%d,%d) feat cols [%d,%d) max|feat diff| %.3e" % (L, r, nlo, nhi, f0, f1, dfeat))
    print("  moments rel diff: sum %.2e sumsq %.2e" % (((sums - sc).abs().max() / sc.abs().max()).item(), ((sqs - ssc).abs().max() / ssc.abs().max()).item()))
    M2, b2, info2 = wct.solve(nc, sums, sqs, ns, ss, sss, 1.0, want_info=True)
    print("  M rel diff %.2e  b rel diff %.2e  sweeps %s %s  |M|max %.3e" % (((M2 - M).abs().max() / M.abs().max()).item(), ((b2 - b).abs().max() / b.abs().max()).item(), info, info2, M.abs().max().item()))
    W_next = (W_cur >> sh) << sh
    for r in range(world):
        o = wct.decode_affine(L, feats[r], M2, b2)
        o_sameM = wct.decode_affine(L, feats[r], M, b)
        own = (owns[r][0], min(owns[r][1], W_next))
        a, bb = own[0] - los[r], own[1] - los[r]
        ref = out_full[..., own[0]:own[1]]
        print("  rank%d owned region: diff with sharded M %.3e, with untiled M %.3e (ref max %.3f)" % (r, (o[..., a:bb] - ref).abs().max().item(), (o_sameM[..., a:bb] - ref).abs().max().item(), ref.abs().max().item()))
        v = CUM_HALO[L] - LEVEL_HALO[L]
        va, vb = max(0, own[0] - v), min(W_next, own[1] + v)
        print("        valid region [%d,%d): diff with untiled M %.3e" % (va, vb, (o_sameM[..., va - los[r]:vb - los[r]] - out_full[..., va:vb]).abs().max().item()))
        imgs[r] = o
        owns[r] = own
    img_full = out_full
    W_cur = W_next
