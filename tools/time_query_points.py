"""query() on random points: time of the nearest-triangle kernel per strategy (ICON_AMD_POINT_SEARCH)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feats = [T(a.features)]; cal = torch.eye(4, device="cuda")[None]
g = torch.Generator(device="cuda"); g.manual_seed(0)
for n in (2000, 20000, 60000, 200000, 1000000, 3000000):
    pts = (torch.rand((1, 3, n), device="cuda", generator=g) * 2 - 1)
    eng.query(feats, pts, cal); eng._work().profile(True)
    acc = 0.0
    for _ in range(5):
        eng.query(feats, pts, cal); acc += eng._work().stage_ms()[0]
    eng._work().profile(False)
    print(f"n={n}: features stage {acc / 5 * 1000:.0f} us")
