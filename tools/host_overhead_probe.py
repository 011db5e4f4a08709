import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import torch
from types import SimpleNamespace
from icon_amd import synth
from icon_amd.engine import IconQueryEngine, query_func
from icon_amd.recon import DenseReconEngine
dev = torch.device('cuda:0')
a = synth.make_assets('body'); T = lambda x: torch.from_numpy(x).to(dev)
eng = IconQueryEngine(prior_type='icon', sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feats = [T(a.features)]
res = int(sys.argv[1]) if len(sys.argv) > 1 else 65
recon = DenseReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[33, res], align_corners=True, engine=eng).to(dev)
opt = SimpleNamespace(num_views=1)
f = lambda: recon(opt=opt, netG=eng, features=feats, proj_matrix=None)
for _ in range(20): f()
torch.cuda.synchronize()
N = 300
t = time.perf_counter()
for _ in range(N): f()
torch.cuda.synchronize()
print(f"res {res}: {(time.perf_counter()-t)/N*1e3:.4f} ms per forward")
# split: eval_slab (async) vs none-check
im = feats[0]
t = time.perf_counter()
for _ in range(N): occ = eng.eval_slab(im, res, 0, res)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"eval_slab host enqueue {(t1-t)/N*1e6:.1f} us per call; drained after {(t2-t)/N*1e3:.4f} ms per call")
occ = eng.eval_slab(im, res, 0, res); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(N): recon._none_if_empty(occ)
print(f"_none_if_empty on an idle stream {(time.perf_counter()-t)/N*1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(N): f()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
# the bench's own instrumentation inside its timed loop: Workspace.stage_ms (event synchronise + three elapsed-time reads)
w = eng._work(); w.profile(True)
f(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(N): w.stage_ms()
print(f"stage_ms on finished events {(time.perf_counter()-t)/N*1e6:.1f} us per call")
t = time.perf_counter()
for _ in range(N): f()
torch.cuda.synchronize()
a_ = (time.perf_counter()-t)/N*1e3
t = time.perf_counter()
for _ in range(N): f(); w.stage_ms()
torch.cuda.synchronize()
b_ = (time.perf_counter()-t)/N*1e3
print(f"forward with events recorded {a_:.4f} ms; + stage_ms every step {b_:.4f} ms")
w.profile(False)
t = time.perf_counter()
for _ in range(N): f()
torch.cuda.synchronize()
print(f"forward without events {(time.perf_counter()-t)/N*1e3:.4f} ms")
