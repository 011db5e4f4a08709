"""Dense (one launch over 257^3) vs the reference's coarse-to-fine schedule on the same fast query."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from icon_amd import synth
from icon_amd.engine import IconQueryEngine, query_func
from icon_amd.recon import DenseReconEngine, AdaptiveReconEngine
a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feats = [T(a.features)]; opt = SimpleNamespace(num_views=1)
kw = dict(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[33, 65, 129, 257], align_corners=True, faster=True)
which = os.environ.get("WHICH", "dense,adaptive,host").split(",")
for name, cls in (("dense", DenseReconEngine), ("adaptive", AdaptiveReconEngine), ("host", AdaptiveReconEngine)):
    if name not in which:
        continue
    rec = cls(**kw).cuda()
    if name == "host":
        rec.native = False                       # the host-driven schedule (torch bookkeeping around HIP queries)
    f = lambda: rec(opt=opt, netG=eng, features=feats, proj_matrix=None)
    f(); f()
    for rep in range(int(os.environ.get("REPEAT", "1"))):       # (the first timed loop of the first process on a box runs slow)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): occ = f()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t) * 100:.3f} ms per volume", getattr(rec, "last_stats", None))

if os.environ.get("PERCALL"):                  # every call on its own (the stream drained in between): where do slow loops come from?
    rec = AdaptiveReconEngine(**kw).cuda()
    f = lambda: rec(opt=opt, netG=eng, features=feats, proj_matrix=None)
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(60):
        t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("per call ms:", " ".join(f"{x:.2f}" for x in ts))

if os.environ.get("PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    rec = AdaptiveReconEngine(**kw).cuda()
    f = lambda: rec(opt=opt, netG=eng, features=feats, proj_matrix=None)
    f(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3): f()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=60))
