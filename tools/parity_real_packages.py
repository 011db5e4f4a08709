#!/usr/bin/env python3
"""ONE command that pins every third-party leaf of the hot path against the REAL package, wherever the packages exist next to a
HIP device (none of them is in this image: every section then prints ABSENT and the script exits 0).

    python tools/parity_real_packages.py [--res 65] [--only kaolin,mcubes,...] [--stand-ins]

For each package: import it if present, run the product leaf (HIP, through the C ABI) and the package on the synthetic subject
exactly as the reference's call site does, and print one PASS / DIFF line with the count and the CLASS of every difference
(tie, ordering, constant offset, winding ...).  Sections and the SURVEY.md section 8 rows they flip from "partial" to "yes":

  kaolin.point_to_mesh_distance / check_sign      lib/dataset/mesh_util.py:374,393     a7 (the far field's tie rule)
  pytorch3d Meshes.verts_normals_padded           lib/dataset/mesh_util.py:367          a7
  kaolin voxelgrids_to_trianglemeshes             lib/common/seg3d_lossless.py:599      f1 (grids <= 256^3)
  PyMCubes marching_cubes                         lib/common/seg3d_lossless.py:592      f1 (grids  > 256^3)
  trimesh split (clean_mesh)                      lib/dataset/mesh_util.py:783          f1
  pytorch3d rasterize_meshes (get_visibility)     lib/dataset/mesh_util.py:298          f3
  voxelize_cuda forward_semantic_voxelization     lib/net/voxelize.py:57                a16 / f4

--stand-ins runs every section with the repo's own CPU checkers (oracle/) in the packages' place: a self-test of this harness
(everything must PASS; needs the HIP device, nothing else).  Exit code: 0 unless --strict and a DIFF was printed."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RESULTS = []


def report(section, status, text):
    RESULTS.append((section, status, text))
    print(f"[{status:6s}] {section}: {text}", flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# mesh comparison: what is different, and of which class
# ---------------------------------------------------------------------------------------------------------------------
def _rows_sorted(a):
    a = np.asarray(a)
    return a[np.lexsort(a.T[::-1])] if len(a) else a


def _edge_keys(v, tol):
    """marching-cubes vertices lie on lattice edges: two coordinates are integers, the third is not.  -> the edge's identity
    ((lower end point, axis) packed into one int64), -1 where the third coordinate is within 2 tol of an integer as well
    (a crossing AT a lattice point: up to six edges' vertices cluster there - those are matched by position among themselves)"""
    r = np.round(v)
    fr = v - r
    ar = np.arange(len(v))
    ax = np.argmax(np.abs(fr), axis=1)
    amb = np.abs(fr[ar, ax]) < 2 * tol
    lo = r.copy()
    lo[ar, ax] = np.floor(v[ar, ax])
    lo = lo.astype(np.int64) + 4
    key = ((lo[:, 0] * 8192 + lo[:, 1]) * 8192 + lo[:, 2]) * 3 + ax
    key[amb] = -1
    return key


def compare_meshes(va, fa, vb, fb, tol=1e-4, lattice=False):
    """(ours, theirs) -> dict: vertex counts, whether the vertex SETS agree (within tol; after a constant offset if one
    aligns them), whether the vertex ORDER agrees, triangle sets as coordinate triples (order- and rotation-free), how many
    triangles of each are missing in the other, how many common triangles are wound the other way round, face order."""
    va, vb = np.asarray(va, np.float64).reshape(-1, 3), np.asarray(vb, np.float64).reshape(-1, 3)
    fa, fb = np.asarray(fa, np.int64).reshape(-1, 3), np.asarray(fb, np.int64).reshape(-1, 3)
    out = {"verts": (len(va), len(vb)), "faces": (len(fa), len(fb)), "offset": None}
    if len(va) == 0 or len(vb) == 0:
        out.update(same_vertex_set=len(va) == len(vb), same_vertex_order=len(va) == len(vb), same_face_set=len(fa) == len(fb),
                   same_face_order=len(fa) == len(fb), only_ours=len(fa), only_theirs=len(fb), flipped=0, max_vertex_diff=None)
        return out
    off = np.zeros(3)
    if len(va) == len(vb):
        d = _rows_sorted(np.round(vb / tol).astype(np.int64)) - _rows_sorted(np.round(va / tol).astype(np.int64))
        cand = np.median(d, axis=0) * tol
        if np.abs(cand).max() > 2 * tol and np.abs(_rows_sorted(np.round((vb - cand) / tol).astype(np.int64))
                                                   - _rows_sorted(np.round(va / tol).astype(np.int64))).max() <= 2:
            off = cand
            out["offset"] = tuple(float(x) for x in np.round(cand, 4))
    vb = vb - off
    # vertices are matched by position (nearest neighbour within tol), triangles compared through the matched ids: no
    # rounding of coordinates into bins, whose edges a 1e-7 difference could straddle
    if va.shape == vb.shape and np.array_equal(va, vb):            # the same array (clean_mesh keeps a subset of its input): by index
        dist, near = np.zeros(len(va)), np.arange(len(va))
    elif lattice:
        # marching-cubes meshes in voxel units: match by the lattice edge a vertex lies on (nearby DISTINCT vertices - several
        # edges crossing close to one lattice point - are closer to each other than float32 resolves at coordinate 500)
        from scipy.spatial import cKDTree
        ka_, kb_ = _edge_keys(va, tol), _edge_keys(vb, tol)
        order = np.argsort(ka_, kind="stable")
        pos = np.searchsorted(ka_[order], kb_)
        pos = np.minimum(pos, len(va) - 1)
        hit = (kb_ >= 0) & (ka_[order][pos] == kb_)
        near = np.where(hit, order[pos], -1)
        dist = np.where(hit, np.abs(va[np.maximum(near, 0)] - vb).max(1), np.inf)
        rest_b = np.flatnonzero(~hit)
        rest_a = np.setdiff1d(np.arange(len(va)), near[hit])
        if len(rest_b) and len(rest_a):                            # the clusters at lattice points: closest pairs first, one to one
            dd, nn = cKDTree(va[rest_a]).query(vb[rest_b], k=min(8, len(rest_a)))
            dd, nn = dd.reshape(len(rest_b), -1), nn.reshape(len(rest_b), -1)
            cand = sorted((dd[i, j], i, nn[i, j]) for i in range(len(rest_b)) for j in range(dd.shape[1]) if np.isfinite(dd[i, j]))
            used_a, used_b = set(), set()
            for d_, i, j in cand:
                if i not in used_b and j not in used_a and d_ <= 4 * tol:
                    used_b.add(i); used_a.add(j)
                    near[rest_b[i]], dist[rest_b[i]] = rest_a[j], d_
        near = np.where(near < 0, 0, near)
    else:
        from scipy.spatial import cKDTree
        dist, near = cKDTree(va).query(vb, k=1)
    matched = dist <= 4 * tol if lattice else dist <= 2 * tol
    ids_b = np.where(matched, near, -1 - np.arange(len(vb)))       # their vertex -> our id (unmatched: a unique negative)
    bijective = bool(matched.all() and len(np.unique(near)) == len(vb) == len(va))
    out["same_vertex_set"] = bijective
    out["same_vertex_order"] = bool(bijective and np.array_equal(near, np.arange(len(va))))
    out["max_vertex_diff"] = float(dist.max()) if bijective else None

    def tri_keys(ids):                                             # rotate the smallest id to the front: keeps the winding
        rot = np.argmin(ids, axis=1)
        return np.stack([np.take_along_axis(ids, ((rot + k) % 3)[:, None], 1)[:, 0] for k in range(3)], 1)
    ka, kb = tri_keys(fa), tri_keys(ids_b[fb])
    sa = {tuple(r) for r in ka.tolist()}
    sb = {tuple(r) for r in kb.tolist()}
    fl_a = {(r[0], r[2], r[1]) for r in sa}
    fl_b = {(r[0], r[2], r[1]) for r in sb}
    ro, rt = sorted(sa - sb - fl_b), sorted(sb - sa - fl_a)
    out["clustered"] = 0
    if ro and rt and len(ro) * len(rt) <= 4_000_000:
        # leftovers whose three corners coincide within 4 tol, winding kept: the same triangle, its vertices matched to the
        # wrong members of a cluster of crossings around one lattice point (values within ~tol of the level)
        pos_b = {}
        for i, near_i in enumerate(ids_b):
            pos_b.setdefault(int(near_i), vb[i])
        def corners(t, theirs):
            return np.stack([(pos_b[int(i)] if theirs else va[int(i)]) for i in t])
        left = [corners(t, True) for t in rt]
        taken = set()
        for t in ro:
            co = corners(t, False)
            for j, ct in enumerate(left):
                if j in taken:
                    continue
                if any(np.abs(co - np.roll(ct, k, axis=0)).max() <= 4 * tol for k in range(3)):
                    taken.add(j); out["clustered"] += 1
                    break
    out["only_ours"] = len(ro) - out["clustered"]
    out["only_theirs"] = len(rt) - out["clustered"]
    fl = [t for t in (sa & fl_b) - sb]
    # (a triangle with two corners at the same place - crossings exactly AT a lattice point - has no winding to compare)
    fl = [t for t in fl if np.linalg.norm(np.cross(va[t[1]] - va[t[0]], va[t[2]] - va[t[0]])) > (4 * tol) ** 2]
    out["flipped"] = len(fl)
    out["same_face_set"] = out["only_ours"] == 0 and out["only_theirs"] == 0 and out["flipped"] == 0
    out["same_face_order"] = bool(len(ka) == len(kb) and np.array_equal(ka, kb))
    return out


def mesh_verdict(section, c, note=""):
    if c["same_vertex_set"] and c["same_face_set"] and c["offset"] is None:
        klass = []
        if not c["same_vertex_order"]:
            klass.append("vertex ORDER differs")
        if not c["same_face_order"]:
            klass.append("face ORDER differs")
        report(section, "PASS", f"same {c['verts'][0]} vertices and {c['faces'][0]} triangles as sets"
               + (f" ({c['clustered']} of them through vertices of a cluster at a lattice point)" if c.get("clustered") else "")
               + (f" (max |dv| {c['max_vertex_diff']:.2e})" if c["max_vertex_diff"] is not None else "")
               + ("; " + ", ".join(klass) if klass else "; same order") + note)
    else:
        report(section, "DIFF", f"verts ours/theirs {c['verts']}, faces {c['faces']}, vertex set equal {c['same_vertex_set']}, constant offset "
               f"{c['offset']}, triangles only ours {c['only_ours']} / only theirs {c['only_theirs']} / wound the other way {c['flipped']}" + note)


# ---------------------------------------------------------------------------------------------------------------------
# the subject and the product leaves
# ---------------------------------------------------------------------------------------------------------------------
class Subject:
    def __init__(self, res, mesh):
        import torch
        from icon_amd import synth, _lib
        from icon_amd.engine import IconQueryEngine, MeshHandle
        _lib.require_device()
        self.torch, self.dev = torch, torch.device("cuda:0")
        self.a = a = synth.make_assets(mesh)
        self.T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(self.dev)
        self.verts, self.faces = self.T(a.smpl_verts), self.T(a.smpl_faces)
        self.handle = MeshHandle(self.verts, self.faces, self.T(a.smpl_cmap), self.T(a.smpl_vis))
        self.eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
        self.eng.set_mesh(self.verts, self.faces, self.T(a.smpl_cmap), self.T(a.smpl_vis))
        self.eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
        self.res = res
        self._vol = {}

    def volume(self, res):
        if res not in self._vol:
            self._vol[res] = self.eng.eval_slab(self.T(self.a.features), res, 0, res).clone()
        return self._vol[res]


def sec_distance_sign(S, point_to_mesh_distance, check_sign, index_vertices_by_faces, name):
    """lib/dataset/mesh_util.py:369-393"""
    from icon_amd import synth
    torch = S.torch
    pts = np.concatenate([synth.lattice_points(S.res), synth.stratified_points(S.a.smpl_verts[0], S.a.smpl_faces[0], 20000, seed=7)])
    p = S.T(pts)
    ours = S.handle.sdf_query(p)
    ties = S.handle.sdf_query_ties(p)
    tri = index_vertices_by_faces(S.verts, S.faces[0])
    d2, idx, _ = point_to_mesh_distance(p[None].contiguous(), tri)
    ins = check_sign(S.verts, S.faces[0], p[None]).reshape(-1).to(S.dev)
    d2, idx = d2.reshape(-1).to(S.dev), idx.reshape(-1).to(S.dev)
    our_d2 = (ours["sdf"].abs() * np.sqrt(3.0)) ** 2
    tied = ties["ulps"] <= 1
    diff = (our_d2 - d2).abs()
    face_diff = ours["face"] != idx
    untied_face = int((face_diff & ~tied).sum())
    d_untied = float(diff[~tied].max()) if (~tied).any() else 0.0
    d_tied = float(diff[tied].max()) if tied.any() else 0.0
    # (our d^2 is rebuilt from the float32 sdf = sqrt(d^2) / sqrt(3): two roundings, a few 1e-7 relative)
    rel = diff / (1.0 + d2)
    ok = untied_face == 0 and float(rel.max()) <= 1e-6
    report(name + " point_to_mesh_distance", "PASS" if ok and int(face_diff.sum()) == 0 else ("PASS" if ok else "DIFF"),
           f"{len(pts)} points, tied fraction {float(tied.float().mean()):.4f}; max |d^2 diff| untied {d_untied:.3e} tied {d_tied:.3e}; nearest face differs "
           f"on {int(face_diff.sum())} points, {int((face_diff & tied).sum())} of them TIES (runner-up within 1 ulp), {untied_face} not"
           + ("" if int(face_diff.sum()) == 0 else " - class: tie rule (ours = exact minimum, lowest index); set IconQueryEngine.tie_rule to match and re-run"))
    sign_diff = int((ours["inside"] != ins.bool()).sum())
    report(name + " check_sign", "PASS" if sign_diff == 0 else "DIFF", f"inside flag differs on {sign_diff} of {len(pts)} points"
           + ("" if sign_diff == 0 else " - class: ray / triangle-edge incidence rule (ours: +x ray, half-open edges, oracle/icon_oracle.c)"))


def sec_vertex_normals(S, verts_normals, name):
    """lib/dataset/mesh_util.py:367"""
    theirs = verts_normals(S.verts, S.faces).reshape(-1, 3).to(S.dev)
    ours = S.handle.vertex_normals().reshape(-1, 3)
    d = float((ours - theirs).abs().max())
    report(name + " verts_normals_padded", "PASS" if d <= 1e-6 else "DIFF", f"max |n - theirs| = {d:.3e} over {ours.shape[0]} vertices"
           + ("" if d <= 1e-6 else " - class: accumulation order of the area-weighted face normals (index_add)"))


def sec_marching_cubes(S, mesher, name, res):
    """lib/common/seg3d_lossless.py:583-604: `mesher(final)` -> (verts, faces) in the reference's output conventions"""
    from icon_amd.recon import DenseReconEngine
    occ = S.volume(res)
    eng = DenseReconEngine(resolutions=[res], align_corners=True)
    vo, fo = eng.export_mesh(occ)
    vt, ft = mesher(occ[1:, 1:, 1:].contiguous())
    c = compare_meshes(vo.numpy(), fo.numpy(), np.asarray(vt), np.asarray(ft), lattice=True)
    mesh_verdict(f"{name} ({res}^3 volume)", c)
    return (vo, fo), (vt, ft)


def sec_clean_mesh(S, split_largest, name):
    """lib/dataset/mesh_util.py:778-791"""
    from icon_amd.recon import clean_mesh, export_mesh_device
    torch = S.torch
    rng = np.random.RandomState(5)
    for label, occ in (("noise 49^3", torch.from_numpy(rng.rand(49, 49, 49).astype(np.float32)).to(S.dev)), (f"body {S.res}^3", S.volume(S.res))):
        v, f = export_mesh_device(occ, 0.5)
        cv, cf = clean_mesh(v, f)
        tv, tf = split_largest(v.cpu().numpy(), f.cpu().numpy())
        c = compare_meshes(cv.cpu().numpy(), cf.cpu().numpy(), tv, tf, tol=1e-5)
        mesh_verdict(f"{name} clean_mesh ({label})", c, "" if c["same_face_order"] else
                     " - face order inside a component is numpy's unstable argsort of the labels upstream (oracle/mc_check.py: largest_component_scipy)")


def sec_visibility(S, their_visibility, name):
    """lib/dataset/mesh_util.py:280-316 with TestDataset.compute_vis_cmap's call pattern"""
    from icon_amd.engine import get_visibility
    torch = S.torch
    verts = S.verts[0]
    xy, z = verts.split([2, 1], dim=1)
    ours = get_visibility(xy, -z, S.faces[0].long()).reshape(-1).cpu()
    theirs = their_visibility(xy, -z, S.faces[0].long()).reshape(-1).cpu()
    nd = int((ours != theirs).sum())
    report(name + " get_visibility", "PASS" if nd == 0 else "DIFF", f"{nd} of {len(ours)} vertices differ (visible ours {int(ours.sum())} / theirs {int(theirs.sum())})"
           + ("" if nd == 0 else " - class: pixel-centre / top-left fill rule or depth ties of the 4096^2 rasteriser (oracle/icon_oracle.c: orc_visibility)"))


def sec_voxelize(S, their_voxelize, name):
    """lib/net/voxelize.py:57-59,119-137"""
    from icon_amd import synth
    from icon_amd.engine import semantic_voxelization
    torch = S.torch
    a = S.a
    vv, tets, code = synth.make_tetra_body(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0])
    ours = semantic_voxelization(S.T(vv)[None], S.T(tets)[None], code, res=128, sigma=0.05)[0].permute(1, 2, 3, 0)
    theirs = their_voxelize(vv, tets, code, 128, 0.05).to(S.dev)
    occ_o, occ_t = ours.abs().sum(-1) > 0, theirs.abs().sum(-1) > 0
    n_occ = int((occ_o != occ_t).sum())
    both = occ_o & occ_t
    d = float((ours - theirs)[both].abs().max()) if both.any() else 0.0
    ok = n_occ == 0 and d <= 1e-4
    report(name + " forward_semantic_voxelization", "PASS" if ok else "DIFF", f"occupied voxels ours {int(occ_o.sum())} / theirs {int(occ_t.sum())}, {n_occ} differ; max |code diff| on the "
           f"common ones {d:.3e}" + ("" if ok else " - class: inside-tetrahedron test on faces (ours: closed, float32, oracle/icon_oracle.c) / Gaussian weight normalisation (1e-3 floor)"))


# ---------------------------------------------------------------------------------------------------------------------
# the packages (or their stand-ins)
# ---------------------------------------------------------------------------------------------------------------------
def load(name, importer):
    try:
        return importer()
    except Exception as e:
        report(name, "ABSENT", f"does not import here ({type(e).__name__}: {e}) - the leaf stays PARITY UNPINNED")
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=65)
    ap.add_argument("--mesh", default="body")
    ap.add_argument("--only", default="")
    ap.add_argument("--stand-ins", action="store_true", help="self-test: the repo's CPU checkers in the packages' place (must all PASS)")
    ap.add_argument("--strict", action="store_true")
    ap.add_argument("--json", default="", help="also write the lines as JSON (section, status, text) to this file")
    args = ap.parse_args()
    only = {s for s in args.only.split(",") if s}
    want = lambda k: not only or k in only
    S = [None]

    def subject():
        if S[0] is None:
            S[0] = Subject(args.res, args.mesh)
        return S[0]

    if args.stand_ins:
        import torch
        from oracle import oracle as orc, mc_check, mc_classic
        tag = "stand-in"

        def p2m(points, tri):
            t = tri[0].cpu().numpy().astype(np.float32)
            v = t.reshape(-1, 3)
            f = np.arange(len(v), dtype=np.int64).reshape(-1, 3)
            d2, idx = orc.Accel(v, f).nearest(points[0].cpu().numpy())
            return torch.from_numpy(d2)[None], torch.from_numpy(idx)[None], None

        def csign(verts, faces, points):
            return torch.from_numpy(orc.check_sign(verts[0].cpu().numpy(), faces.cpu().numpy(), points[0].cpu().numpy()))[None]
        ivf = lambda verts, faces: verts[:, faces.long()]
        if want("kaolin"):
            sec_distance_sign(subject(), p2m, csign, ivf, tag)
        if want("pytorch3d"):
            sec_vertex_normals(subject(), lambda v, f: torch.from_numpy(orc.vertex_normals(v[0].cpu().numpy(), f[0].cpu().numpy())), tag)
            sec_visibility(subject(), lambda xy, z, f: torch.from_numpy(orc.visibility(xy.cpu().numpy(), z.cpu().numpy().reshape(-1), f.cpu().numpy(), 4096)), tag)

        def classic(final):
            v, f = mc_classic.marching_cubes(final.cpu().numpy(), 0.5)
            return v[:, [2, 1, 0]], f[:, [0, 2, 1]]
        if want("mcubes"):
            sec_marching_cubes(subject(), classic, tag + " classic marching cubes", min(args.res, 65))
        if want("trimesh"):
            sec_clean_mesh(subject(), lambda v, f: mc_check.largest_component_scipy(v, f)[:2], tag)
        if want("voxelize_cuda"):
            sec_voxelize(subject(), lambda vv, tets, code, res, sigma: torch.from_numpy(orc.semantic_voxelize(vv, len(code), code, tets, res=res, sigma=sigma)), tag)
    else:
        # ---- kaolin -----------------------------------------------------------------------------------------------
        if want("kaolin"):
            def imp():
                import kaolin
                from kaolin.metrics.trianglemesh import point_to_mesh_distance
                from kaolin.ops.mesh import check_sign, index_vertices_by_faces
                from kaolin.ops.conversions import voxelgrids_to_trianglemeshes
                return kaolin, point_to_mesh_distance, check_sign, index_vertices_by_faces, voxelgrids_to_trianglemeshes
            k = load("kaolin", imp)
            if k:
                name = f"kaolin {k[0].__version__}"
                sec_distance_sign(subject(), k[1], k[2], k[3], name)

                def kaolin_mc(final):                                    # seg3d_lossless.py:597-602
                    vertices, triangles = k[4](final.unsqueeze(0))
                    return vertices[0][:, [2, 1, 0]].cpu().numpy(), triangles[0][:, [0, 2, 1]].cpu().numpy()
                sec_marching_cubes(subject(), kaolin_mc, name + " voxelgrids_to_trianglemeshes", args.res)
        # ---- PyMCubes ---------------------------------------------------------------------------------------------
        if want("mcubes"):
            def imp():
                import mcubes
                return mcubes
            m = load("PyMCubes", imp)
            if m:
                def pymcubes(final):                                     # seg3d_lossless.py:587-596
                    vertices, triangles = m.marching_cubes(final.detach().cpu().numpy(), 0.5)
                    return vertices[:, [2, 1, 0]], triangles.astype(np.int64)[:, [0, 2, 1]]
                sec_marching_cubes(subject(), pymcubes, "PyMCubes marching_cubes", args.res)
        # ---- trimesh ----------------------------------------------------------------------------------------------
        if want("trimesh"):
            def imp():
                import trimesh
                return trimesh
            t = load("trimesh", imp)
            if t:
                def split_largest(v, f):                                 # mesh_util.py:782-786
                    lst = t.Trimesh(v, f).split(only_watertight=False)
                    comp_num = [m_.vertices.shape[0] for m_ in lst]
                    best = lst[comp_num.index(max(comp_num))]
                    return np.asarray(best.vertices, np.float32), np.asarray(best.faces, np.int32)
                sec_clean_mesh(subject(), split_largest, f"trimesh {t.__version__}")
        # ---- pytorch3d --------------------------------------------------------------------------------------------
        if want("pytorch3d"):
            def imp():
                import pytorch3d
                from pytorch3d.structures import Meshes
                from pytorch3d.renderer.mesh import rasterize_meshes
                return pytorch3d, Meshes, rasterize_meshes
            p3 = load("pytorch3d", imp)
            if p3:
                import torch
                name = f"pytorch3d {p3[0].__version__}"
                sec_vertex_normals(subject(), lambda v, f: p3[1](v, f).verts_normals_padded(), name)

                def their_vis(xy, z, faces):                             # mesh_util.py:280-316 with the settings of render_utils.py:178-186
                    xyz = (torch.cat((xy, -z), dim=1) + 1.0) / 2.0
                    meshes = p3[1](verts=xyz[None], faces=faces[None])
                    pix_to_face, _, _, _ = p3[2](meshes, image_size=2 ** 12, blur_radius=0.0, faces_per_pixel=1, bin_size=None,
                                                 max_faces_per_bin=None, perspective_correct=True, cull_backfaces=True)
                    ids = torch.unique(faces[torch.unique(pix_to_face), :])
                    vis = torch.zeros(z.shape[0])
                    vis[ids.cpu()] = 1.0
                    return vis
                sec_visibility(subject(), their_vis, name)
        # ---- voxelize_cuda ----------------------------------------------------------------------------------------
        if want("voxelize_cuda"):
            def imp():
                import voxelize_cuda
                return voxelize_cuda
            vc = load("voxelize_cuda", imp)
            if vc:
                import torch

                def their_vox(vv, tets, code, res, sigma):               # voxelize.py:40-59 as Voxelization.forward prepares the operands
                    dev = torch.device("cuda:0")
                    v = torch.from_numpy(vv)[None].to(dev)
                    tet = v.reshape(-1, 3)[torch.from_numpy(tets).long().to(dev)][None].contiguous()
                    n_s = len(code)
                    occ = torch.zeros((1, res, res, res), device=dev)
                    sem = torch.zeros((1, res, res, res, 3), device=dev)
                    wsum = torch.full((1, res, res, res), 1e-3, device=dev)
                    occ, sem, wsum = vc.forward_semantic_voxelization(v[:, :n_s].contiguous(), torch.from_numpy(np.asarray(code, np.float32))[None].to(dev).contiguous(),
                                                                      tet, occ, sem, wsum, sigma)
                    return sem[0]
                sec_voxelize(subject(), their_vox, "voxelize_cuda")
    n_diff = sum(1 for r in RESULTS if r[1] == "DIFF")
    n_pass = sum(1 for r in RESULTS if r[1] == "PASS")
    n_abs = sum(1 for r in RESULTS if r[1] == "ABSENT")
    if args.json:
        import json
        with open(args.json, "w") as f:
            json.dump({"pass": n_pass, "diff": n_diff, "absent": n_abs,
                       "lines": [{"section": a, "status": b, "text": c} for a, b, c in RESULTS]}, f, indent=1)
    print(f"parity_real_packages: {n_pass} PASS, {n_diff} DIFF, {n_abs} ABSENT")
    return 1 if (args.strict and n_diff) else 0


if __name__ == "__main__":
    sys.exit(main())
