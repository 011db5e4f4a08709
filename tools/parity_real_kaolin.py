"""Optional real-asset run (last row): the HIP leaves against the REAL kaolin leaves, where kaolin is
installed next to a HIP device (it is not in this image - the script then says so and exits 0).

    python tools/parity_real_kaolin.py [--res 65] [--mesh body]

Compares, on the lattice of the given resolution plus a near-surface sample:
  kaolin.metrics.trianglemesh.point_to_mesh_distance  (lib/dataset/mesh_util.py:374)  vs  icon_sdf_query d^2 / face
  kaolin.ops.mesh.check_sign                          (lib/dataset/mesh_util.py:393)  vs  icon_sdf_query inside
and prints the three numbers bench.py's parity object carries - max |d^2 difference| on untied / tied points and the
tied fraction - plus how many nearest-face choices differ and whether every difference is a tie (runner-up within 1 ulp).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=65)
    ap.add_argument("--mesh", default="body")
    args = ap.parse_args()
    try:
        import kaolin
        from kaolin.metrics.trianglemesh import point_to_mesh_distance
        from kaolin.ops.mesh import check_sign, index_vertices_by_faces
    except Exception as e:
        print(f"parity_real_kaolin: kaolin does not import here ({e!r}) - nothing to compare; the leaves stay PARITY UNPINNED "
              "(DESIGN.md section 2).  Run this where kaolin 0.11.0 and a HIP/CUDA build of torch coexist.")
        return 0
    import torch
    from icon_amd import synth
    from icon_amd.engine import MeshHandle
    dev = torch.device("cuda:0")
    a = synth.make_assets(args.mesh)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    verts, faces = T(a.smpl_verts), T(a.smpl_faces)
    pts = np.concatenate([synth.lattice_points(args.res), synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], 20000, seed=7)])
    p = T(pts)
    h = MeshHandle(verts, faces, T(a.smpl_cmap), T(a.smpl_vis))
    ours = h.sdf_query(p)
    ties = h.sdf_query_ties(p)
    tri = index_vertices_by_faces(verts, faces[0])                        # what face_vertices() builds, mesh_util.py:369
    d2, idx, _ = point_to_mesh_distance(p[None].contiguous(), tri)
    ins = check_sign(verts, faces[0], p[None]).reshape(-1)
    d2, idx = d2.reshape(-1), idx.reshape(-1)
    our_d2 = (ours["sdf"].abs() * np.sqrt(3.0)) ** 2
    tied = ties["ulps"] <= 1
    diff = (our_d2 - d2).abs()
    face_diff = ours["face"] != idx
    print(f"kaolin {kaolin.__version__}: {len(pts)} points, frac_tied {tied.float().mean().item():.4f}")
    print(f"  max |d^2 - kaolin| untied {diff[~tied].max().item():.3e}   tied {diff[tied].max().item():.3e}")
    print(f"  nearest face differs on {int(face_diff.sum())} points; of those tied (runner-up within 1 ulp): {int((face_diff & tied).sum())}")
    print(f"  inside flag differs on {int((ours['inside'] != ins.bool()).sum())} points")
    return 0


if __name__ == "__main__":
    sys.exit(main())
