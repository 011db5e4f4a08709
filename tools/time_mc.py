import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
from icon_amd.recon import export_mesh_device, export_mesh_numpy
dev = torch.device("cuda:0"); a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).to(dev)
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis)); eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
occ = eng.eval_slab(T(a.features), 257, 0, 257)
export_mesh_device(occ); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): v, f = export_mesh_device(occ); vc, fc = v.cpu(), f.cpu()
torch.cuda.synchronize(); print("device MC + mesh D2H, ms:", (time.perf_counter() - t0) / 5 * 1e3, v.shape, f.shape)
t0 = time.perf_counter(); h = occ.cpu().numpy(); t1 = time.perf_counter(); vh, fh = export_mesh_numpy(h); t2 = time.perf_counter()
print("volume D2H ms:", (t1 - t0) * 1e3, "host MC ms:", (t2 - t1) * 1e3, vh.shape, fh.shape)
