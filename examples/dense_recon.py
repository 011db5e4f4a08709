#!/usr/bin/env python
"""End-to-end on the synthetic subject: what apps/ICON.py:test_single does after netG.filter() (lines 729-761) -
reconEngine -> export_mesh -> clean_mesh -> vertices into the [-1,1] cube - through the HIP path, with timings.

    python examples/dense_recon.py [--res 257] [--adaptive] [--out body.obj]

Needs an MI355X (there is no CPU path).  The inputs stand in for what the reference computes upstream of the hot path:
`features` = HGPIFuNet.filter() output, the SMPL tensors = TestDataset.compute_vis_cmap(), the regressor = netG.if_regressor.
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=257)
    ap.add_argument("--adaptive", action="store_true", help="the reference's coarse-to-fine schedule instead of the dense lattice")
    ap.add_argument("--out", default=None, help="write the mesh as Wavefront OBJ")
    args = ap.parse_args()
    import torch
    from icon_amd import synth
    from icon_amd.engine import IconQueryEngine, query_func
    from icon_amd.recon import AdaptiveReconEngine, DenseReconEngine, clean_mesh

    dev = torch.device("cuda:0")
    a = synth.make_assets("body")
    T = lambda x: torch.from_numpy(x).to(dev)
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)               # or IconQueryEngine.attach(netG) on a reference network
    eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
    features = [T(a.features)]
    res = args.res
    levels = [r for r in (33, 65, 129, 257, 513) if r <= res] if res in (65, 129, 257, 513) else [res]
    cls = AdaptiveReconEngine if args.adaptive else DenseReconEngine
    recon = cls(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=levels, align_corners=True,
                balance_value=0.5, faster=True).to(dev)
    opt = SimpleNamespace(num_views=1)

    def sync():
        torch.cuda.synchronize(); return time.perf_counter()
    w = recon(opt=opt, netG=eng, features=features, proj_matrix=None)          # warm-up: BVH build, operand packing, workspaces,
    clean_mesh(*recon.export_mesh(w))                                           # ... and the marching-cubes / clean_mesh scratch
    t0 = sync()
    sdf = recon(opt=opt, netG=eng, features=features, proj_matrix=None)
    t1 = sync()
    verts, faces = recon.export_mesh(sdf)                                       # marching cubes on the GPU, CPU tensors out (as upstream)
    t2 = sync()
    verts, faces = clean_mesh(verts, faces)                                     # largest component (apps/ICON.py:755-756)
    t3 = sync()
    half = (res - 1) / 2.0
    verts = (verts.float() - half) / half                                       # apps/ICON.py:758-759
    print(f"{'adaptive' if args.adaptive else 'dense'} {res}^3: volume {1e3 * (t1 - t0):.2f} ms, marching cubes {1e3 * (t2 - t1):.2f} ms, "
          f"clean_mesh {1e3 * (t3 - t2):.2f} ms -> {verts.shape[0]} vertices, {faces.shape[0]} faces, "
          f"bbox {verts.min(0).values.tolist()} .. {verts.max(0).values.tolist()}")
    if args.out:
        v, f = verts.cpu().numpy(), faces.cpu().numpy() + 1
        with open(args.out, "w") as fh:
            fh.writelines(f"v {x:.6f} {y:.6f} {z:.6f}\n" for x, y, z in v)
            fh.writelines(f"f {i} {j} {k}\n" for i, j, k in f)
        print("wrote", args.out)


if __name__ == "__main__":
    main()
