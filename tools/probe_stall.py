"""Where does the ONE slow call (35-55 ms) among the first ~10 native schedules of a process come from?"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feat = T(a.features)
mode = os.environ.get("MODE", "default")
if mode == "nogc": gc.disable()
ts = []
for i in range(30):
    t = time.perf_counter()
    if mode == "nosync":
        eng.adaptive_eval(feat, [33, 65, 129, 257], counts=False)
        t1 = time.perf_counter(); torch.cuda.synchronize()
        ts.append(((t1 - t) * 1e3, (time.perf_counter() - t) * 1e3))
    elif mode == "dense":
        eng.eval_slab(feat, 65, 0, 65); torch.cuda.synchronize(); ts.append(((time.perf_counter() - t) * 1e3,))
    else:
        eng.adaptive_eval(feat, [33, 65, 129, 257]); ts.append(((time.perf_counter() - t) * 1e3,))
print(mode, " ".join("/".join(f"{x:.1f}" for x in t) for t in ts))
