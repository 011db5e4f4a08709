"""per-image preparation cost in steady state: icon_mesh_create (normals, packed records, BVH), icon_feat_create (plane repack),
icon_mlp_create (fold + pack; once per checkpoint) - python tools/prep_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from icon_amd import synth
from icon_amd.engine import MeshHandle, FeatHandle, MlpHandle

a = synth.make_assets("body")
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
v, f, c, vis, feat = T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis), T(a.features)
sd = {k: torch.from_numpy(x) for k, x in a.state_dict.items()}


def tm(fn, n=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); h = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return np.median(ts), min(ts)


rs = np.random.RandomState(0)
print("icon_mesh_create (new vertices every call): median %.2f ms, min %.2f ms" % tm(lambda: MeshHandle(v + 1e-4 * torch.randn_like(v), f, c, vis)))
print("icon_feat_create: median %.2f ms, min %.2f ms" % tm(lambda: FeatHandle(feat, 2)))
print("icon_mlp_create: median %.2f ms, min %.2f ms" % tm(lambda: MlpHandle(sd)))
