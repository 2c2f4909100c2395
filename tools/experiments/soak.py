"""GPU box: a soak of the stylise call -- N frames back to back at config-2 size and at varying sizes, two lanes overlapping throughout --
checking what a benchmark's 20 steps cannot: that the single-launch C = 128 solves never abort into the Jacobi net (wct_debug_get
nscoop_aborts / nscoop_off), that nothing saturates, that the result of the same frame stays bit-identical, that memory does not grow.
usage: python tools/experiments/soak.py [frames=1500]  -> gpurun_out/soak.txt"""
import hashlib, os, sys, time, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import numpy as np, torch
from wct_hip import WCT, model_zoo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
g = torch.Generator(device="cuda").manual_seed(1)
c = torch.rand((1, 3, 2160, 3840), device="cuda", generator=g)
s = torch.rand((1, 3, 2048, 2048), device="cuda", generator=g)
out = torch.empty((3, 2160, 3840), device="cuda")
sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
eng.stylize(c, s, out=out); torch.cuda.synchronize()
h0 = sha(out)
free0 = torch.cuda.mem_get_info()[0]
rng = np.random.default_rng(0)
t0 = time.time()
bad = 0
for i in range(n):
    if i % 10 == 9:      # every tenth frame another size (workspaces regrow only when a size exceeds everything seen)
        H, W = int(rng.integers(64, 1400)), int(rng.integers(64, 2200))
        cc = torch.rand((1, 3, H, W), device="cuda", generator=g)
        r = eng.stylize(cc, s[:, :, :1024, :1024].contiguous())
        if not bool(torch.isfinite(r).all()):
            bad += 1
    else:
        eng.stylize(c, s, out=out)
    if i % 250 == 249:
        torch.cuda.synchronize()
        print("frame %d: %.1f s, sha %s (%s), aborts %d off %d solves %d" % (i + 1, time.time() - t0, sha(out), "same" if sha(out) == h0 else "DIFFERENT",
              eng.debug_get("nscoop_aborts"), eng.debug_get("nscoop_off"), eng.debug_get("nscoop_solves")), flush=True)
eng.stylize(c, s, out=out); torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
line = ("soak: %d frames in %.1f s; config-2 frame sha %s -> %s (%s); non-finite varying-size results %d; saturation count %d; single-launch solves %d, aborted %d, "
        "lanes off %d; device memory free %.2f -> %.2f GiB" % (n, time.time() - t0, h0, sha(out), "bit-identical" if sha(out) == h0 else "DIFFERENT", bad,
        eng.saturation_count(), eng.debug_get("nscoop_solves"), eng.debug_get("nscoop_aborts"), eng.debug_get("nscoop_off"), free0 / 2**30, free1 / 2**30))
print(line)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "soak.txt"), "w").write(line + "\n")
