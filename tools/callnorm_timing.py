"""wall time of one whole-lattice evaluation with a GroupNorm / InstanceNorm regressor (icon_amd/callnorm.py) beside the
BatchNorm (folded, fused) path: python tools/callnorm_timing.py [res]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine

res = int(sys.argv[1]) if len(sys.argv) > 1 else 257
a = synth.make_assets("body")
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
feat = T(a.features)
rs = np.random.RandomState(2)
for kind in ("batch", "group", "instance"):
    sd = dict(a.state_dict) if kind == "batch" else {k: v for k, v in a.state_dict.items() if k.startswith("filters.")}
    if kind == "group":
        for l, c in enumerate((512, 256, 128)):
            sd[f"norms.{l}.weight"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
            sd[f"norms.{l}.bias"] = rs.normal(0, 0.1, c).astype(np.float32)
    eng.norm_mlp = None if kind == "batch" else kind
    eng.set_regressor({k: torch.from_numpy(v) for k, v in sd.items()})
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        occ = eng.eval_slab(feat, res, 0, res)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{kind:9s} {res}^3: {dt * 1e3:8.1f} ms   (max memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB)  inside {int((occ > 0.5).sum())}")
