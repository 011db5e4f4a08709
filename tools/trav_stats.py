import sys; sys.path.insert(0,'.')
import torch
from icon_amd import synth
from icon_amd.engine import MeshHandle
a = synth.make_assets("body"); T=lambda x: torch.from_numpy(x).cuda()
mesh = MeshHandle(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
print(mesh.traversal_stats(257))
