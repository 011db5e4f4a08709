"""Stress: schedule -> marching cubes -> clean_mesh, many times; the cleaned mesh must be the same every time (the hash-table
inserts and the unions of icon_clean_mesh race in a different order on every call)."""
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
n = int(os.environ.get("N", "1000"))
g = torch.Generator(device="cuda"); g.manual_seed(1)
noise = (torch.rand((65, 65, 65), device="cuda", generator=g))          # hundreds of components, pinch vertices
first = None
t0 = time.perf_counter()
for it in range(n):
    vol, _, _ = eng.adaptive_eval(feat, [33, 65, 129, 257], counts=False)
    v, f = export_mesh_device(vol)
    cv, cf = clean_mesh(v, f)
    nv, nf = export_mesh_device(noise)
    ncv, ncf = clean_mesh(nv, nf)
    cur = (cv, cf, ncv, ncf)
    if first is None:
        first = tuple(t.clone() for t in cur)
    elif not all(torch.equal(x, y) for x, y in zip(cur, first)):
        print("MISMATCH at iteration", it, [tuple(t.shape) for t in cur]); sys.exit(1)
torch.cuda.synchronize()
print(f"stress ok: {n} x (schedule + MC + clean_mesh on the body, MC + clean_mesh on noise), {(time.perf_counter() - t0) / n * 1e3:.3f} ms each; "
      f"body {tuple(first[0].shape)} {tuple(first[1].shape)}, noise {tuple(first[2].shape)} {tuple(first[3].shape)}")
