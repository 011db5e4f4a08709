"""Time one dense 257^3 evaluation (engine.eval_slab) without looking at the result: used by the timing-experiment
builds of tools/exp_fused.sh, whose outputs are wrong by construction.  usage: time_fused.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
a = synth.make_assets("body")
T = lambda x: torch.from_numpy(x).to(dev)
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feat = T(a.features)
out = torch.empty((257, 257, 257), device=dev)
for _ in range(2): eng.eval_slab(feat, 257, 0, 257, out=out)
eng._work().profile(True)
st = np.zeros(3)
torch.cuda.synchronize()
for _ in range(steps):
    eng.eval_slab(feat, 257, 0, 257, out=out)
    st += np.array(eng._work().stage_ms())
st /= steps
print(f"{os.environ.get('ICON_AMD_LIB', 'baseline').split('/')[-1]:40s} pre {st[0]:.3f}  mid {st[1]:.3f}  fused/mlp {st[2]:.3f} ms")
