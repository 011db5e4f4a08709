"""Where the wall time of one bench step goes beyond the kernels: back-to-back eval_slab calls (no host sync between
volumes) vs DenseReconEngine.forward (which returns None for an empty volume and therefore synchronises)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine, query_func
from icon_amd.recon import DenseReconEngine
dev = torch.device("cuda:0")
a = synth.make_assets("body")
T = lambda x: torch.from_numpy(x).to(dev)
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feat = T(a.features)
out = torch.empty((257, 257, 257), device=dev)
recon = DenseReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[33, 65, 129, 257],
                         align_corners=True, balance_value=0.5, faster=True, engine=eng).to(dev)
opt = SimpleNamespace(num_views=1)
K = 20
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / K * 1e3
print(f"eval_slab back to back          {timeit(lambda: eng.eval_slab(feat, 257, 0, 257, out=out)):.3f} ms / volume")
def synced():
    eng.eval_slab(feat, 257, 0, 257, out=out); torch.cuda.synchronize()
print(f"eval_slab + synchronize         {timeit(synced):.3f} ms / volume")
print(f"DenseReconEngine.forward        {timeit(lambda: recon(opt=opt, netG=eng, features=[feat], proj_matrix=None)):.3f} ms / volume")
eng._work().profile(True)
tr = ts = 0.0
for i in range(23):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = recon(opt=opt, netG=eng, features=[feat], proj_matrix=None); t1 = time.perf_counter()
    eng._work().stage_ms(); t2 = time.perf_counter()
    if i >= 3: tr += t1 - t0; ts += t2 - t1
print(f"with events: forward {tr / 20 * 1e3:.3f} ms, stage_ms() {ts / 20 * 1e3:.3f} ms")
