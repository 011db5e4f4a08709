"""Stress: fresh meshes + the native schedule, many times; every volume must equal the first one bit for bit (the shared-walk
search and the device mesh build are timing dependent in HOW they get there, never in the answer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
n = int(os.environ.get("N", "300"))
v, f, cm, vs = T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis)
feat = T(a.features)
first = None
t0 = time.perf_counter()
for it in range(n):
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip) if it % 50 == 0 else eng
    if it % 50 == 0:
        eng.set_regressor({k: torch.from_numpy(x) for k, x in a.state_dict.items()})
    eng.set_mesh(v.clone(), f, cm, vs)            # a new tensor: a new device build every time
    vol, counts, pos = eng.adaptive_eval(feat, [33, 65, 129, 257])
    if first is None:
        first, c0 = vol.clone(), counts
    elif not (torch.equal(vol.view(torch.int32), first.view(torch.int32)) and counts == c0):
        print("MISMATCH at iteration", it, counts, c0, int((vol != first).sum())); sys.exit(1)
torch.cuda.synchronize()
print(f"stress ok: {n} meshes + schedules, {(time.perf_counter() - t0) / n * 1e3:.3f} ms each, counts {c0}")
