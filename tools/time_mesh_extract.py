"""The mesh side of an image in the reference's own mode: schedule -> marching cubes -> clean_mesh (apps/ICON.py:729-761)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
from icon_amd.recon import export_mesh_device, clean_mesh
a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feat = T(a.features)
def sync(): torch.cuda.synchronize()
# SCHEDULE="33,65,129,257,513": the shipped default mcube_res=512 (configs/icon-filter.yaml:23)
for sched in ([33, 65, 129, 257], [33, 65, 129, 257, 513]):
  if os.environ.get("ONLY") and int(os.environ["ONLY"]) != sched[-1]:
      continue
  for rep in range(int(os.environ.get("REPEAT", "3"))):
    ts = [0.0, 0.0, 0.0]
    n = 10
    for _ in range(n):
        sync(); t0 = time.perf_counter()
        vol, counts, pos = eng.adaptive_eval(feat, sched); sync(); t1 = time.perf_counter()
        v, f = export_mesh_device(vol); sync(); t2 = time.perf_counter()
        vc, fc = clean_mesh(v, f); sync(); t3 = time.perf_counter()
        ts[0] += t1 - t0; ts[1] += t2 - t1; ts[2] += t3 - t2
    print(f"{sched[-1]}^3: schedule {ts[0] / n * 1e3:.3f} ms  marching cubes {ts[1] / n * 1e3:.3f} ms ({tuple(v.shape)}, {tuple(f.shape)})  "
          f"clean_mesh {ts[2] / n * 1e3:.3f} ms ({tuple(vc.shape)}, {tuple(fc.shape)})")
