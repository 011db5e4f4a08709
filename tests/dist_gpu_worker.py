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
            for overlap, balance, reserve, layout, gather_to in ((True, True, 0, "ab", None), (True, True, 0, "contiguous", None), (False, False, 8, "ab", None),
                                                                 (True, False, 8, "ab", world - 1)):
                def engine():
                    e = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip, cmap_mode=cmap_mode)
                    e.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
                    e.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
                    return e
                eng = engine()
                rec = DenseReconEngine(query_func=query_func, resolutions=[res], align_corners=True, engine=eng, shard=True,
                                       overlap_gather=overlap, balance_slabs=balance, reserve_cus=reserve, slab_layout=layout,
                                       gather_to=gather_to).to(dev)
                vol = rec(opt=opt, netG=eng, features=feats, proj_matrix=None)
                slabs = rec.last_stats["slabs"]
                ab = layout == "ab" and overlap                 # (without the overlapped gather there is one slab per rank, one collective)
                assert rec.last_stats.get("layout", "contiguous") == ("ab" if ab else "contiguous"), rec.last_stats
                if ab:
                    # two slabs per rank, each volume gather straight into its block of the result: no assembly copy
                    assert rec.last_stats["pieces"] == DenseReconEngine.ab_pieces(res, world)[2] and rec.last_stats["assembly_copies"] == 0
                    assert (vol is None) == (gather_to is not None and gather_to != rank)
                else:
                    assert len(slabs) == world and slabs[0][0] == 0 and slabs[-1][1] == res
                    # overlapped gather + the tiled cmap rule: phase 1 ran per half-slab on two workspaces with its own sign exchange
                    assert bool(rec.last_stats.get("split_features")) == (overlap and cmap_mode == "reference"), rec.last_stats
                holder = 0 if gather_to is None else gather_to
                if gather_to is not None and world > 1:          # the destination's volume travels to rank 0 for the comparison
                    if rank == holder:
                        dist.send(vol.cpu().contiguous(), dst=0)
                    if rank == 0:
                        vol = torch.empty((res, res, res))
                        dist.recv(vol, src=holder)
                        vol = vol.to(dev)
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
