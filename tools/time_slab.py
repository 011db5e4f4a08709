"""Per-rank cost of the Z-slab path on ONE GPU (no collectives): what a rank of an N-GPU job computes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from icon_amd import synth
from icon_amd.engine import IconQueryEngine
from icon_amd.recon import slab_bounds

a = synth.make_assets("body"); T = lambda x: torch.from_numpy(x).cuda()
eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
feat = T(a.features); res = 257
for world in (1, 2, 4, 8):
    for rank in range(world):
        z0, z1, per = slab_bounds(res, world, rank)
        out = torch.zeros((per, res, res), device="cuda")
        def step():
            signs, count = eng.slab_features(feat, res, z0, z1)
            k = int(count.item())                                   # the host sync the exchange needs
            eng.slab_finish(res, z0, z1, signs[:k].contiguous(), k, 0, out=out[: z1 - z0], device=out.device)
        step(); torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        ms = sum(ts) / len(ts)
        if max(ts) > 1.5 * min(ts): print("   per-step ms:", [round(x, 2) for x in ts])
        eng._work().profile(True); step(); st = eng._work().stage_ms(); eng._work().profile(False)
        print(f"world {world} rank {rank}: planes {z1 - z0}, {ms:.3f} ms per step stages {[round(x, 2) for x in st]}")
