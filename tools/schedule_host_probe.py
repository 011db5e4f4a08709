import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import torch
from types import SimpleNamespace
from icon_amd import synth
from icon_amd.engine import IconQueryEngine, query_func
from icon_amd.recon import AdaptiveReconEngine
dev = torch.device('cuda:0')
a = synth.make_assets('body'); T = lambda x: torch.from_numpy(x).to(dev)
eng = IconQueryEngine(prior_type='icon', sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feats = [T(a.features)]
ad = AdaptiveReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[33, 65, 129, 257], align_corners=True, faster=True).to(dev)
opt = SimpleNamespace(num_views=1)
f = lambda: ad(opt=opt, netG=eng, features=feats, proj_matrix=None)
g = lambda: eng.adaptive_eval(feats[0], [33, 65, 129, 257])
for _ in range(20): f(); g()
import gc; gc.collect(); gc.freeze(); gc.disable()
import numpy as np
for name, fn in (("engine.adaptive_eval", g), ("AdaptiveReconEngine.forward", f)):
    ts = []
    for _ in range(300):
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
    print(name, "p50 %.4f ms" % np.median(ts))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): f()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
