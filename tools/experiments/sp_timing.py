"""Phase breakdown of conv3x3_sp_kernel per layer of the 4K level-4 encoder (library built by sp_timing.sh, WCT_LIB_PATH)."""
import ctypes, os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT, model_zoo, lib
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
L = lib.load()
c = torch.rand((3, 2160, 3840), device="cuda")
buf = (ctypes.c_ulonglong * 4)()
prev = None
print("DMA-staged convolutions of encoder level k (cumulative over the level's layers): cycles per job (one 16-channel chunk of a unit), wave 0")
for k in (2, 3, 4, 5):
    for _ in range(2):
        wct.encode(k, c, layout="nhwc")
    torch.cuda.synchronize(); L.wct_debug_sp_timing(buf)
    for _ in range(3):
        wct.encode(k, c, layout="nhwc")
    torch.cuda.synchronize(); L.wct_debug_sp_timing(buf)
    t = list(buf)
    jobs = max(t[3], 1)
    print("  encoder %d: %9d jobs | wait+barrier %6.0f | tap loop %6.0f | rest %6.0f   (MFMA issue alone: 9 taps x CT x 2 x 3 x 32 cycles x 2 waves/SIMD = 6912 at CT = 2)"
          % (k, jobs // 3, t[0] / jobs, t[1] / jobs, t[2] / jobs))
