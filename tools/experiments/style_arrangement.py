"""Which style-side arrangement for which rank count?  One rank's share of config 4 (10240x4096 in N strips, 2048x2048 style) through the library's
cascade (wct_stylize_sharded on a 1-rank RCCL communicator, geometry emulated: debug key shard_emulate), style_mode owner against strips, for
N = 2, 4, 8 and the ranks that differ most (0: edge strip + level 5 under owner; N // 2: interior).  ms per frame of 4 back-to-back frames."""
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
os.environ["WCT_DEBUG"] = "1"
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from tests.fixture_compare import noise_frame  # noqa: E402
from wct_hip import WCT, model_zoo  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
eng.comm_init(dist)
H, W = 4096, 10240
frame = torch.from_numpy(noise_frame(5, H, W)).cuda()
style = torch.from_numpy(noise_frame(2, 2048, 2048)).cuda()
eng.style_prepare(style)
for world in (2, 4, 8):
    for r in sorted({0, world // 2, world - 1}):
        row = []
        for smode in ("owner", "strips", "replicate"):
            eng.debug_set("shard_emulate", 100 * world + r)
            own0, own1, in0, in1, mode = eng.shard_geometry(W, world, r, "auto")
            strip = frame[:, :, in0:in1].contiguous()
            for _ in range(2):
                eng.stylize_sharded(strip, style, W, in0, in1, halo_mode="auto", style_mode=smode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                eng.stylize_sharded(strip, style, W, in0, in1, halo_mode="auto", style_mode=smode)
            torch.cuda.synchronize()
            row.append("%s %.3f" % (smode, (time.perf_counter() - t0) / 4 * 1e3))
            eng.debug_set("shard_emulate", 0)
            del strip
        print("N=%d rank %d halo=%s cols_in=%d: %s" % (world, r, mode, in1 - in0, " | ".join(row)), flush=True)
eng.comm_destroy()
dist.destroy_process_group()
