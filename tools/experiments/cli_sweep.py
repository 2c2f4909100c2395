import sys, os, types, json
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"collaborative-distillation_amd")]
import bench, torch
from wct_hip import WCT, model_zoo
w = model_zoo.load_npz_weights("collaborative-distillation_amd/weights/16x.npz")
eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
for depth, io in ((3,8),(4,16),(6,24),(4,32)):
    r = bench.cli_folder_pass(eng, depth=depth, io_threads=io)
    print(depth, io, json.dumps({k:(v.get("pipelined") if isinstance(v,dict) else v) for k,v in r.items() if k!="workload"}), flush=True)
