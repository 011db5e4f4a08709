import sys; sys.path.insert(0,'.')
import torch
from icon_amd import synth
from icon_amd.engine import MeshHandle
a = synth.make_assets("body"); T=lambda x: torch.from_numpy(x).cuda()
mesh = MeshHandle(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
for res in (33, 65, 129, 257):
    print(res, mesh.traversal_stats(res))
import time
for n in (36000, 100000, 1000000):
    pts = (torch.rand((n, 3), device="cuda") * 2 - 1)
    mesh.sdf_query(pts); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): mesh.sdf_query(pts)
    torch.cuda.synchronize(); print(f"sdf_query {n} random points: {(time.perf_counter()-t)*200:.3f} ms")
