"""Where a rank frame of the 8-GPU config-4 job goes: per-family kernel times (overlap off, HIP events around every launch) of rank 3's share through
wct_stylize_sharded (1-rank RCCL communicator, geometry emulated) beside ONE EIGHTH of the untiled 10240x4096 frame's family times on the same GPU."""
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
os.environ["WCT_DEBUG"] = "1"
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from tests.fixture_compare import noise_frame  # noqa: E402
from wct_hip import WCT, model_zoo  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
eng.comm_init(dist)
H, W, world, r = 4096, 10240, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 3
frame = torch.from_numpy(noise_frame(5, H, W)).cuda()
style = torch.from_numpy(noise_frame(2, 2048, 2048)).cuda()
N = 3


def families(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    eng.set_overlap(False)
    eng.profile_reset()
    eng.profile(True)
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    eng.profile(False)
    eng.set_overlap(True)
    return {e["name"]: (e["ms"] / N, e["launches"] // N) for e in eng.profile_read()}


out_full = torch.empty((3, H, W), device="cuda")
full = families(lambda: eng.stylize(frame, style, out=out_full))
del out_full
eng.style_prepare(style)
eng.debug_set("shard_emulate", 100 * world + r)
own0, own1, in0, in1, mode = eng.shard_geometry(W, world, r, "auto")
strip = frame[:, :, in0:in1].contiguous()
rank = families(lambda: eng.stylize_sharded(strip, style, W, in0, in1, halo_mode="auto", style_mode="owner", fast_fold=True))
eng.debug_set("shard_emulate", 0)
names = sorted(set(full) | set(rank), key=lambda n: -(rank.get(n, (0, 0))[0]))
print("rank %d of %d (%d columns in, %d owned; halo %s), kernel time per frame, overlap off" % (r, world, in1 - in0, own1 - own0, mode))
print("%-40s %10s %6s %14s %8s" % ("family", "rank ms", "n", "untiled/8 ms", "ratio"))
tr = tf = 0.0
for n in names:
    a, b = rank.get(n, (0.0, 0)), full.get(n, (0.0, 0))
    tr += a[0]
    tf += b[0] / world
    print("%-40s %10.4f %6d %14.4f %8s" % (n, a[0], a[1], b[0] / world, ("%.2f" % (a[0] / (b[0] / world))) if b[0] > 0 else "-"))
print("%-40s %10.4f %6s %14.4f %8.2f" % ("SUM", tr, "", tf, tr / tf))
eng.comm_destroy()
dist.destroy_process_group()
