"""Random geometries through the library's sharded cascade against the Python orchestration, bitwise (rank threads of one process over device-buffer
collectives, tools/sharded_standins.py): rank count, frame size (odd sizes, widths that floor pooling shrinks), style size, halo / style / map arrangement."""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tools import sharded_standins as standins  # noqa: E402
from wct_hip import WCT, model_zoo  # noqa: E402
from wct_hip.sharded import ShardedStylizer  # noqa: E402

w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)   # noqa: E731
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for case in range(n):
    world = int(rng.integers(2, 9))
    halo = str(rng.choice(["recompute", "exchange", "auto"]))
    smode = str(rng.choice(["owner", "strips", "replicate", "auto"]))
    bmap = bool(rng.integers(0, 2))
    alpha = float(rng.choice([1.0, 0.6]))
    wmin = 16 * world + 16 if halo != "exchange" else 144 * world + 32
    W = int(rng.integers(wmin, wmin + 1200))
    H = int(rng.integers(32, 200))
    Hs, Ws = int(rng.integers(32, 160)), int(rng.integers(max(32, 16 * world + 8), 400))
    try:
        ShardedStylizer(None, None, H, W, Hs, Ws, rank=0, world=world, halo_mode=halo, style_mode=smode)
    except ValueError:
        continue
    g = torch.Generator(device="cuda").manual_seed(case)
    content = torch.rand((3, H, W), device="cuda", generator=g)
    style = torch.rand((3, Hs, Ws), device="cuda", generator=g)
    kw = dict(halo_mode=halo, broadcast_map=bmap, style_mode=smode, alpha=alpha)
    want, gp = standins.run_in_process(world, make, content, style, **kw)
    got, gc = standins.run_in_process(world, make, content, style, c_cascade=True, **kw)
    same = bool(torch.equal(got, want)) and [x.calls for x in gc] == [x.calls for x in gp]
    ref = make().stylize(content, style, alpha=alpha)
    err = float((got - ref).abs().max() / ref.abs().max())
    print("case %2d world %d %4dx%-4d style %3dx%-3d halo %-9s style_mode %-9s bmap %d alpha %.1f: bitwise %s, vs untiled %.2e" % (
        case, world, W, H, Ws, Hs, halo, smode, bmap, alpha, same, err), flush=True)
    bad += (not same) or not (err < 2e-3)
print("failures: %d" % bad)
