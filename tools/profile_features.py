"""Stage-level timing of the feature kernel on the 257^3 lattice (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine, MeshHandle

dev = torch.device("cuda:0")
a = synth.make_assets("body")
T = lambda x: torch.from_numpy(x).to(dev)
res = int(sys.argv[1]) if len(sys.argv) > 1 else 257
mesh = MeshHandle(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
print("mesh stats", mesh.stats())

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

st = mesh.traversal_stats(res)
print("traversal stats", st, "nodes/pt(wave-amortised)", st["nodes_per_wave"] / 64, "tris/pt", st["tris_per_wave"] / 64)
print("traversal-only kernel ms (incl. malloc/memcpy overhead)", timed(lambda: mesh.traversal_stats(res)))
for prec in ("f16x3",):
    for mode in ("reference", "local"):
        eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip, cmap_mode=mode, precision=prec)
        eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
        eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
        feat = T(a.features)
        eng.eval_slab(feat, res, 0, res); eng._work().profile(True)
        acc = np.zeros(3)
        for _ in range(3):
            eng.eval_slab(feat, res, 0, res); acc += np.array(eng._work().stage_ms())
        print(mode, prec, "stage ms (features, patch, mlp)", acc / 3)
pts = T(synth.lattice_points(res, res // 2 - 8, res // 2 + 8))
print("sdf_query point-mode on 16 planes (x-fastest order), ms:", timed(lambda: mesh.sdf_query(pts)), "points", len(pts))
