"""Is wct_stylize_sharded -- RCCL calls included -- capturable into ONE HIP graph?  One-rank RCCL communicator, the geometry of rank 3 of an 8-rank
config-4 job emulated (debug key shard_emulate): five ncclAllReduce, the owner-mode broadcast of the owned level, four grouped ncclSend / ncclRecv.
Prints the host time per frame of the direct call and of graph.replay(), and whether the replay is bitwise the direct call."""
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

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
eng.comm_init(dist)
H, W, world, r = 4096, 10240, 8, 3
style = torch.from_numpy(noise_frame(2, 2048, 2048)).cuda()
eng.style_prepare(style)
eng.debug_set("shard_emulate", 100 * world + r)
own0, own1, in0, in1, mode = eng.shard_geometry(W, world, r, "auto")
strip = torch.from_numpy(noise_frame(5, H, W)[:, :, in0:in1].copy()).cuda()
out = torch.empty(3 * H * (own1 - own0), device="cuda")
for smode in ("owner", "strips"):
    for _ in range(3):
        want = eng.stylize_sharded(strip, style, W, in0, in1, halo_mode="auto", style_mode=smode, out=out).clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        eng.stylize_sharded(strip, style, W, in0, in1, halo_mode="auto", style_mode=smode, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s direct: host %.3f ms per frame, frame %.3f ms" % (smode, (t1 - t0) / 8 * 1e3, (t2 - t0) / 8 * 1e3), flush=True)
    try:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                eng.stylize_sharded(strip, style, W, in0, in1, halo_mode="auto", style_mode=smode, out=out)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(out.view(-1)[:want.numel()], want.view(-1)))
        t0 = time.perf_counter()
        for _ in range(8):
            graph.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s graph: capture ok, replay bitwise %s, host %.3f ms per frame, frame %.3f ms" % (smode, same, (t1 - t0) / 8 * 1e3, (t2 - t0) / 8 * 1e3), flush=True)
    except Exception as e:      # noqa: BLE001
        print("%s graph: capture FAILED: %r" % (smode, e), flush=True)
        break
