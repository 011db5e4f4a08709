"""One dense slab call on the coarse lattices (33^3 .. 129^3): how the search's packet size / wave sharing choices time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feat = T(a.features)
out = []
for res in (33, 65, 129):
    for _ in range(3): eng.eval_slab(feat, res, 0, res)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): eng.eval_slab(feat, res, 0, res)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) * 100)
    out.append(f"{res}^3 {best:.3f} ms")
print("slab:", "  ".join(out))
