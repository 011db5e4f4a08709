import sys, time; sys.path.insert(0,'.')
import torch
from icon_amd import synth
from icon_amd.engine import MeshHandle
a = synth.make_assets("body"); T=lambda x: torch.from_numpy(x).cuda()
mesh = MeshHandle(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
g = torch.Generator(device="cuda"); g.manual_seed(0)
for n in (64, 1024, 8192, 36000, 200000, 1000000, 1999999, 2000001):
    pts = (torch.rand((n, 3), device="cuda", generator=g) * 2 - 1)
    mesh.sdf_query(pts); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20): mesh.sdf_query(pts)
    ev1.record(); torch.cuda.synchronize()
    print(f"sdf_query {n} random points: {ev0.elapsed_time(ev1)/20*1000:.1f} us (GPU time incl. output allocs)")
near = torch.tensor(a.smpl_verts[0][:64]).cuda() + 0.01
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
mesh.sdf_query(near); ev0.record()
for _ in range(20): mesh.sdf_query(near)
ev1.record(); torch.cuda.synchronize(); print(f"64 near-surface points: {ev0.elapsed_time(ev1)/20*1000:.1f} us")
