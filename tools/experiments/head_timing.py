"""Phase breakdown of enc_head_kernel on the 4K content (library built by head_timing.sh, loaded through WCT_LIB_PATH)."""
import ctypes, os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "collaborative-distillation_amd")]
import torch
from wct_hip import WCT, model_zoo, lib
w = model_zoo.load_npz_weights(os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
L = lib.load()
c = torch.rand((3, 2160, 3840), device="cuda")
buf = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    wct.encode(2, c, layout="nhwc")
torch.cuda.synchronize()
L.wct_debug_head_timing(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    wct.encode(2, c, layout="nhwc")   # enc_head + conv21
e1.record(); torch.cuda.synchronize()
L.wct_debug_head_timing(buf)
t = list(buf)
names = ["wait barrier 1", "issue next fetch", "conv11 + split + store", "barrier 2", "conv12 MFMA (c16_compute)", "pool epilogue + store", "commit next image"]
tot = sum(t[:7])
print("encode(2) %.3f ms per call; %d workgroup-runs" % (e0.elapsed_time(e1) / 5, t[7]))
for n, v in zip(names, t[:7]):
    print("  %-28s %5.1f %%   %8.0f cycles per tile" % (n, 100.0 * v / tot, v / 5 / 32400 ))
