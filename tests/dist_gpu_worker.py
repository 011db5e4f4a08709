"""Worker of tests/test_gpu_ties_shell.py::test_real_ranks_on_one_gpu_run_the_hip_slab_kernels: started once per rank by
torch.distributed.run (gloo: several ranks may share the one GPU of the test box).  Every rank drives the REAL HIP slab
kernels of its Z-slab through DenseReconEngine's sharded protocol (packed sign messages, pieces, gathered finish); rank 0
compares the assembled volume with the single-process evaluation of the same engine settings - bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from types import SimpleNamespace
    from icon_amd import synth
    from icon_amd.engine import IconQueryEngine, query_func
    from icon_amd.recon import DenseReconEngine
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    a = synth.make_assets("body")
    T = lambda x: torch.from_numpy(x).to(dev)
    opt = SimpleNamespace(num_views=1)
    feats = [T(a.features)]
    checked = 0
    for res in (65, 129):
        for cmap_mode in ("reference", "local"):
            for overlap, balance, reserve in ((True, True, 0), (False, False, 8)):
                def engine():
                    e = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip, cmap_mode=cmap_mode)
                    e.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
                    e.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
                    return e
                eng = engine()
                rec = DenseReconEngine(query_func=query_func, resolutions=[res], align_corners=True, engine=eng, shard=True,
                                       overlap_gather=overlap, balance_slabs=balance, reserve_cus=reserve).to(dev)
                vol = rec(opt=opt, netG=eng, features=feats, proj_matrix=None)
                slabs = rec.last_stats["slabs"]
                assert len(slabs) == world and slabs[0][0] == 0 and slabs[-1][1] == res
                # overlapped gather + the tiled cmap rule: phase 1 ran per half-slab on two workspaces with its own sign exchange
                assert bool(rec.last_stats.get("split_features")) == (overlap and cmap_mode == "reference"), rec.last_stats
                if rank == 0:
                    e1 = engine()
                    one = DenseReconEngine(query_func=query_func, resolutions=[res], align_corners=True, engine=e1, shard=False).to(dev)
                    ref = one(opt=opt, netG=e1, features=feats, proj_matrix=None)
                    assert vol.shape == ref.shape == (res, res, res)
                    assert torch.equal(vol, ref), (res, cmap_mode, overlap, float((vol - ref).abs().max()))
                    checked += 1
                dist.barrier()
    if rank == 0:
        print(f"DIST_GPU_OK world={world} checked={checked}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
