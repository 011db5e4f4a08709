import os, sys, time
sys.path.insert(0, '.')
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
from icon_amd.recon import slab_bounds
a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feat = T(a.features); res = 257
def tm(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for (world, rank) in ((4, 1), (4, 2), (4, 2), (4, 3)):
    z0, z1, per = slab_bounds(res, world, rank)
    out = torch.zeros((per, res, res), device="cuda")
    for it in range(3):
        (signs, count), t1 = tm(lambda: eng.slab_features(feat, res, z0, z1))
        k, t2 = tm(lambda: int(count.item()))
        sg, t3 = tm(lambda: signs[:k].contiguous())
        _, t4 = tm(lambda: eng.slab_finish(res, z0, z1, sg, k, 0, out=out[: z1 - z0], device=out.device))
        print(f"world {world} rank {rank} z {z0}-{z1} it {it}: features {t1:.2f} item {t2:.2f} slice {t3:.2f} finish {t4:.2f}  k={k} signs.numel={signs.numel()}")
