"""CPU only: the reference arithmetic with torch CPU fp32 convolutions (oneDNN) vs the oracle (C loops) vs their fp64 arms,
end to end on the 5-level 16x cascade -- how far apart are two VALID fp32 implementations of the reference?
usage: python tools/experiments/torch_vs_oracle.py 1080 1920 1024 1024   (measured: see tests/test_hip_scale.py)"""
import os, sys, time
REPO="/root/repo"; sys.path[:0]=[REPO, REPO+"/collaborative-distillation_amd"]
import numpy as np, torch, torch.nn.functional as F
from oracle import wct_oracle
from wct_hip import model_zoo
torch.set_num_threads(8)
w = model_zoo.load_npz_weights(REPO+"/collaborative-distillation_amd/weights/16x.npz")
H,W,Hs,Ws = (int(v) for v in sys.argv[1:5])
rng = np.random.default_rng(0)
c = rng.random((3,H,W),dtype=np.float32); s = rng.random((3,Hs,Ws),dtype=np.float32)
class TorchMods:
    def __init__(self, dt): self.dt=dt; self.precision="fp64" if dt==torch.float64 else "fp32"
    def conv(self,x,wt,b): return F.relu(F.conv2d(F.pad(x,(1,1,1,1),mode="reflect"), torch.from_numpy(wt).to(self.dt), torch.from_numpy(b).to(self.dt)))
    def encode(self, level, img):
        key="e%d"%level
        y=torch.from_numpy(np.asarray(img)).to(self.dt)[None]
        y=F.conv2d(y, torch.from_numpy(w[key+".conv0.weight"]).to(self.dt), torch.from_numpy(w[key+".conv0.bias"]).to(self.dt))
        for l in model_zoo.encoder_layers("16x", level):
            y=self.conv(y,w["%s.%s.weight"%(key,l.name)],w["%s.%s.bias"%(key,l.name)])
            if l.pool_after: y=F.max_pool2d(y,2,2)
        return y[0].numpy()
    def decode(self, level, feat):
        key="d%d"%level
        y=torch.from_numpy(np.asarray(feat)).to(self.dt)[None]
        for l in model_zoo.decoder_layers("16x", level):
            y=self.conv(y,w["%s.%s.weight"%(key,l.name)],w["%s.%s.bias"%(key,l.name)])
            if l.up_after: y=F.interpolate(y,scale_factor=2,mode="nearest")
        return y[0].numpy()
rel=lambda a,b: float(np.abs(a-b).max()/np.abs(b).max())
t=time.time()
r_t32 = wct_oracle.stylize(TorchMods(torch.float32), c, s, 1.0)
r_t64 = wct_oracle.stylize(TorchMods(torch.float64), c, s, 1.0)
print("torch", time.time()-t)
wct_oracle.set_num_threads(8)
r_o32 = wct_oracle.stylize(wct_oracle.Modules("16x", w), c, s, 1.0)
r_o64 = wct_oracle.stylize(wct_oracle.Modules("16x", w, precision="fp64"), c, s, 1.0)
print("torch32 vs torch64 %.3e | oracle32 vs oracle64 %.3e | torch64 vs oracle64 %.3e | torch32 vs oracle32 %.3e | torch32 vs oracle64 %.3e" % (rel(r_t32,r_t64), rel(r_o32,r_o64), rel(r_t64,r_o64), rel(r_t32,r_o32), rel(r_t32, r_o64)))
