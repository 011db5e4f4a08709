"""Worker of tests/test_gpu_ties_shell.py::test_real_ranks_exchange_meshes_instead_of_the_volume: several ranks (gloo; they share
the test box's one GPU) run DenseReconEngine.forward_mesh - every rank triangulates the cell layers of its own Z-slab, keyed
vertices and faces are gathered and merged; every rank compares with marching cubes on the single-process volume: the same
vertices and faces in the same order."""
import os
import sys

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
    for res in (33, 65, 129):
        for cmap_mode in ("reference", "local"):
            for balance in (True, False):
                def engine():
                    e = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip, cmap_mode=cmap_mode)
                    e.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
                    e.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
                    return e
                eng = engine()
                rec = DenseReconEngine(query_func=query_func, resolutions=[res], align_corners=True, engine=eng, shard=True,
                                       balance_slabs=balance).to(dev)
                out = rec.forward_mesh(opt=opt, netG=eng, features=feats, proj_matrix=None)
                assert rec.last_stats.get("gather") == "mesh", rec.last_stats
                e1 = engine()
                one = DenseReconEngine(query_func=query_func, resolutions=[res], align_corners=True, engine=e1, shard=False).to(dev)
                ref = one.export_mesh(one(opt=opt, netG=e1, features=feats, proj_matrix=None))
                assert out is not None and out[0].shape == ref[0].shape and out[1].shape == ref[1].shape, (res, out[0].shape, ref[0].shape, out[1].shape, ref[1].shape)
                assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), (res, cmap_mode, balance)
                assert rec.last_stats["exchanged_bytes"] < 0.35 * world * res ** 3 * 4 or res < 65       # well under the volume's bytes
                checked += 1
                dist.barrier()
    # nothing above 0.5 anywhere: None, like forward()
    e0 = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
    e0.set_mesh(T(a.smpl_verts * 0.02 + 5.0), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))     # a body far outside the cube
    e0.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
    rec = DenseReconEngine(query_func=query_func, resolutions=[33], align_corners=True, engine=e0, shard=True).to(dev)
    one = DenseReconEngine(query_func=query_func, resolutions=[33], align_corners=True, engine=e0, shard=False).to(dev)
    want_none = one(opt=opt, netG=e0, features=feats, proj_matrix=None) is None
    got = rec.forward_mesh(opt=opt, netG=e0, features=feats, proj_matrix=None)
    assert (got is None) == want_none, (want_none, got is None)
    if rank == 0:
        print(f"DIST_MESH_OK world={world} checked={checked} none={want_none}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
