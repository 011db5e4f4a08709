import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icon_amd import synth
from icon_amd.engine import MeshHandle, FeatHandle, MlpHandle
dev = torch.device("cuda:0"); a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).to(dev)
args = (T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis)); f = T(a.features)
sd = {k: torch.from_numpy(v) for k, v in a.state_dict.items()}
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m = MeshHandle(*args); torch.cuda.synchronize(); t1 = time.perf_counter()
    fh = FeatHandle(f, 2); torch.cuda.synchronize(); t2 = time.perf_counter(); mh = MlpHandle(sd); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"mesh {1e3*(t1-t0):.2f} ms  feat {1e3*(t2-t1):.2f} ms  mlp {1e3*(t3-t2):.2f} ms")
