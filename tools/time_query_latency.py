"""Wall-clock latency of IconQueryEngine.query for small batches (host overhead + kernels)."""
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
for n in (1000, 1, 1000, 3000, 1000, 8000, 50000):
    pts = (torch.rand((1, 3, n), device="cuda", generator=g) * 2 - 1)
    for _ in range(3): eng.query(feats, pts, cal)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): out = eng.query(feats, pts, cal)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t) / 50 * 1e6
    eng._work().profile(True); eng.query(feats, pts, cal); st = eng._work().stage_ms(); eng._work().profile(False)
    print(f"n={n}: wall {wall:.0f} us per call; GPU stages (features, patch, mlp) us: {[round(x * 1000) for x in st]}")
