"""Marching cubes against the independent checker (oracle/mc_check.py), clean_mesh and display.

The reference's own mesh extraction is third-party code that is absent here (kaolin / PyMCubes,
lib/common/seg3d_lossless.py:583-604): PARITY UNPINNED for the triangulation.  What is pinned: the vertex set
(table-independent), one-cube triangles, closed + consistently oriented surface, outward normals, Euler
characteristic; clean_mesh against a numpy restatement of lib/dataset/mesh_util.py:778-791; display()
against the reference's own output (tests/golden/display_33.npz, made by running Seg3dLossless.display
verbatim - tools/make_golden.py section g).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from common import ROOT, assets, golden
from icon_amd.recon import DenseReconEngine, export_mesh_numpy
from oracle import mc_check


def _sphere(res, r=0.6):
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, res)] * 3, indexing="ij")
    return (0.5 + (r - np.sqrt(x * x + y * y + z * z))).astype(np.float32)


def _check(occ, v, f, closed_genus0=None):
    v, f = np.asarray(v, np.float64), np.asarray(f, np.int64)
    assert mc_check.same_point_set(v, mc_check.edge_crossings(occ, 0.5)), "vertex set differs from the edge crossings"
    t = mc_check.topology(v, f, occ.shape[0] - 1)
    assert t["one_cube"] and t["used_all"] and t["oriented"] and t["closed"], t
    if closed_genus0 is not None:
        assert t["watertight"] and t["components"] == closed_genus0 and t["euler"] == 2 * closed_genus0, t
        assert t["signed_volume"] > 0, t          # outward normals under export_mesh's winding flip
    return t


def test_host_marching_cubes_vs_independent_checker():
    occ = _sphere(33)
    v, f = export_mesh_numpy(occ, 0.5)
    t = _check(occ, v.numpy(), f.numpy(), closed_genus0=1)
    r = 0.6 * 16
    assert abs(t["signed_volume"] - 4 / 3 * np.pi * r ** 3) / (4 / 3 * np.pi * r ** 3) < 0.02
    body = golden("seg3d_body_dense33.npz")["occ"]                 # the reference's own dense volume
    v, f = export_mesh_numpy(body, 0.5)
    _check(body, v.numpy(), f.numpy())
    noise = np.random.RandomState(0).rand(17, 17, 17).astype(np.float32)     # every ambiguous case, open at the border
    v, f = export_mesh_numpy(noise, 0.5)
    _check(noise, v.numpy(), f.numpy())


def test_display_matches_reference_output():
    g = golden("display_33.npz")
    eng = DenseReconEngine(resolutions=[17, 33], align_corners=True)
    img = eng.display(torch.from_numpy(g["vol"]))
    assert img.dtype == np.uint8 and img.shape == g["image"].shape
    assert np.array_equal(img, g["image"])


@pytest.mark.gpu
def test_device_marching_cubes_257_vs_independent_checker():
    from icon_amd.recon import export_mesh_device
    from test_gpu_parity import make_engine, T
    a = assets("body")
    occ = make_engine(a).eval_slab(T(a.features), 257, 0, 257)
    v, f = export_mesh_device(occ, 0.5)
    t = _check(occ.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy())
    assert t["watertight"] and t["signed_volume"] > 0
    # closed orientable surfaces: chi = sum(2 - 2 g_i) is even and at most 2 per component (the clipped-sdf
    # field of the synthetic checkpoint has small handles / blobs around the clip band, so genus > 0 occurs)
    assert t["euler"] % 2 == 0 and t["euler"] <= 2 * t["components"]
    print("257^3 mesh:", t)


def test_merge_of_keyed_mesh_pieces_restores_the_whole_mesh():
    """recon.merge_keyed_meshes (the last step of the sharded mesh exchange, DenseReconEngine.forward_mesh) on the CPU: a host
    marching-cubes mesh with its vertices keyed by their order, cut into 3 'ranks' by face ranges - every piece re-indexes its
    own vertices, so vertices used on both sides of a cut appear in two pieces - merges back into exactly the original"""
    from icon_amd.recon import export_mesh_numpy, merge_keyed_meshes
    z, y, x = np.meshgrid(*([np.linspace(-1, 1, 33)] * 3), indexing="ij")
    occ = (0.7 - np.sqrt(x * x + 0.8 * y * y + 1.3 * z * z)).astype(np.float32) + 0.5
    v, f = export_mesh_numpy(occ, 0.5)
    assert f.shape[0] > 1000
    keys = torch.arange(v.shape[0], dtype=torch.int64) * 7 + 3              # any strictly increasing keys
    cuts = [0, f.shape[0] // 3, f.shape[0] // 2, f.shape[0]]
    pk, pv, pf = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        used = torch.unique(f[a:b].reshape(-1))                              # ascending: a piece's vertices keep their relative order
        local = torch.full((v.shape[0],), -1, dtype=torch.int64)
        local[used] = torch.arange(used.shape[0])
        pk.append(keys[used]); pv.append(v[used]); pf.append(local[f[a:b]])
    assert sum(k.shape[0] for k in pk) > v.shape[0]                          # the cuts do duplicate vertices
    mv, mf = merge_keyed_meshes(pk, pv, pf)
    assert torch.equal(mv, v.float()) and torch.equal(mf, f.long())


def test_checker_face_connectivity_vs_vertex_connectivity():
    """oracle/mc_check.py: largest_component_by_faces (trimesh's edge-based face adjacency, the checker of icon_clean_mesh)
    against largest_component (vertex connectivity): equal on closed surfaces without pinch vertices; at a pinch (two
    tetrahedra sharing ONE vertex) the edge-based rule keeps them apart"""
    from icon_amd import synth
    v1, f1 = synth.icosphere(2, radius=0.5)
    v2, f2 = synth.icosphere(1, radius=0.2, center=(2.0, 0.0, 0.0))
    v = np.concatenate([v1, v2]).astype(np.float32)
    f = np.concatenate([f2 + len(v1), f1])                         # the small component listed first
    a, b = mc_check.largest_component_by_faces(v, f), mc_check.largest_component(v, f)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and len(a[0]) == len(v1)
    t = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]], np.int64)
    vp = np.random.RandomState(0).rand(7, 3).astype(np.float32)
    fp = np.concatenate([t, np.array([0, 4, 5, 6])[t]])            # vertex 0 is in both tetrahedra
    pv, pf = mc_check.largest_component_by_faces(vp, fp)
    assert len(pv) == 4 and len(pf) == 4 and np.array_equal(pv, vp[:4])    # equal sizes: the one holding face 0
    wv, wf = mc_check.largest_component(vp, fp)
    assert len(wv) == 7 and len(wf) == 8                           # a vertex union-find merges them


def _rows_sorted(f):
    f = np.asarray(f)
    return f[np.lexsort(f.T[::-1])]


@pytest.mark.parametrize("case", ["noise", "body", "two_spheres_small_first", "pinch", "equal_sizes"])
def test_checker_component_rule_vs_the_scipy_engine_trimesh_calls(case):
    """lib/dataset/mesh_util.py:778-791 = trimesh.split + first-largest.  trimesh is absent; the engine it calls for the
    components - scipy.sparse.csgraph.connected_components on the face-adjacency graph - is not.  mc_check.largest_component_scipy
    restates trimesh's bookkeeping around that engine step by step; the plain-Python checker of icon_clean_mesh
    (largest_component_by_faces) must pick the SAME component: same vertices in the same order, same face set.  What this
    also shows: the ORDER of the faces inside trimesh's submesh is the order numpy's default (unstable) argsort leaves the
    label groups in - not ascending in general (noise case) - so only the face SET of clean_mesh is comparable with upstream."""
    from icon_amd import synth
    if case == "noise":                                             # 1,000+ components, pinch vertices, open at the border
        v, f = export_mesh_numpy(np.random.RandomState(3).rand(33, 33, 33).astype(np.float32), 0.5)
        v, f = v.numpy(), f.numpy()
    elif case == "body":
        v, f = export_mesh_numpy(golden("seg3d_body_dense33.npz")["occ"], 0.5)
        v, f = v.numpy(), f.numpy()
    elif case == "two_spheres_small_first":
        v1, f1 = synth.icosphere(2, radius=0.5)
        v2, f2 = synth.icosphere(1, radius=0.2, center=(2.0, 0.0, 0.0))
        v, f = np.concatenate([v1, v2]).astype(np.float32), np.concatenate([f2 + len(v1), f1])
    elif case == "pinch":                                           # two tetrahedra sharing one vertex: two components
        t = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]], np.int64)
        v = np.random.RandomState(0).rand(7, 3).astype(np.float32)
        f = np.concatenate([t, np.array([0, 4, 5, 6])[t]])
    else:                                                           # equal sizes: the FIRST of several largest (comp_num.index(max))
        v1, f1 = synth.icosphere(1, radius=0.3)
        v = np.concatenate([v1, v1 + 2.0, v1 - 2.0]).astype(np.float32)
        f = np.concatenate([f1 + len(v1), f1, f1 + 2 * len(v1)])   # face 0 belongs to the middle copy
    cv, cf = mc_check.largest_component_by_faces(v, f)
    sv, sf, ascending = mc_check.largest_component_scipy(v, f)
    assert np.array_equal(cv, sv)
    assert cf.shape == sf.shape and np.array_equal(_rows_sorted(cf), _rows_sorted(sf))
    if ascending:
        assert np.array_equal(cf, sf)
    if case == "equal_sizes":
        assert np.array_equal(cv, v[len(v1): 2 * len(v1)])
    if case == "noise":
        assert len(cv) > 1000 and len(cv) < len(v)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["noise", "body"])
def test_clean_mesh_vs_the_scipy_engine(case):
    """icon_clean_mesh itself against the scipy-engine restatement of trimesh's rule (same image on the GPU box)"""
    from icon_amd.recon import clean_mesh, export_mesh_device
    from test_gpu_parity import make_engine, T
    dev = torch.device("cuda:0")
    if case == "noise":
        occ = torch.from_numpy(np.random.RandomState(5).rand(49, 49, 49).astype(np.float32)).to(dev)
    else:
        a = assets("body")
        occ = make_engine(a).eval_slab(T(a.features), 129, 0, 129)
    v, f = export_mesh_device(occ, 0.5)
    cv, cf = clean_mesh(v, f)
    sv, sf, _ = mc_check.largest_component_scipy(v.cpu().numpy(), f.cpu().numpy())
    assert np.array_equal(cv.cpu().numpy(), sv)
    assert np.array_equal(_rows_sorted(cf.cpu().numpy()), _rows_sorted(sf))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["noise", "body"])
def test_clean_mesh_keeps_the_largest_component(case):
    from icon_amd.recon import clean_mesh, export_mesh_device, mesh_components
    from test_gpu_parity import make_engine, T
    dev = torch.device("cuda:0")
    if case == "noise":
        occ = torch.from_numpy(np.random.RandomState(3).rand(33, 33, 33).astype(np.float32)).to(dev)
    else:
        a = assets("body")
        occ = make_engine(a).eval_slab(T(a.features), 65, 0, 65)
    v, f = export_mesh_device(occ, 0.5)
    ev, ef = mc_check.largest_component(v.cpu().numpy(), f.cpu().numpy())
    cv, cf = clean_mesh(v, f)
    assert cv.is_cuda and cv.dtype == torch.float32 and cf.dtype == torch.int32
    assert np.array_equal(cv.cpu().numpy(), ev) and np.array_equal(cf.cpu().numpy(), ef)
    cv2, cf2 = clean_mesh(v.cpu(), f.cpu())                        # the reference passes CPU tensors (export_mesh output)
    assert not cv2.is_cuda and np.array_equal(cv2.numpy(), ev) and np.array_equal(cf2.numpy(), ef)
    lab = mesh_components(f, v.shape[0]).cpu().numpy()
    assert len(np.unique(lab[np.unique(f.cpu().numpy())])) == mc_check.topology(v.cpu().numpy(), f.cpu().numpy(), occ.shape[0] - 1)["components"]


@pytest.mark.gpu
def test_display_on_device_volume():
    g = golden("display_33.npz")
    eng = DenseReconEngine(resolutions=[17, 33], align_corners=True)
    assert np.array_equal(eng.display(torch.from_numpy(g["vol"]).cuda()), g["image"])


@pytest.mark.gpu
def test_chamfer_p2s_on_the_hip_closest_point_engine():
    """icon_amd.metrics.chamfer_p2s (lib/dataset/Evaluator.py:200-230 on the exact nearest-triangle kernel): zero for a
    mesh against itself, equal to the checker-based restatement (tests/common.py: chamfer) within sampling noise for
    two different surfaces, and p2s == the gt-samples -> prediction half."""
    from common import chamfer
    from icon_amd import metrics, synth
    dev = torch.device("cuda:0")
    v1, f1 = synth.icosphere(3, radius=0.6)
    v2, f2 = synth.icosphere(3, radius=0.63, center=(0.05, -0.02, 0.01))
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    c0, p0 = metrics.chamfer_p2s(T(v1), T(f1), T(v1), T(f1), n=100_000)
    assert c0 <= 1e-3 and p0 <= 1e-3                                  # x100 scale: 1e-5 in cube units
    c, p = metrics.chamfer_p2s(T(v2), T(f2), T(v1), T(f1), n=100_000)
    c_ref, _ = chamfer(v2, f2, v1, f1, n=20000)
    assert abs(c - c_ref) <= 0.03 * c_ref + 0.02, (c, c_ref)
    assert 0.5 * c <= p <= 2.0 * c
    assert 2.0 <= c <= 6.0                                            # radii differ by 0.03, centres by 0.055 -> a few x100 units


# ---------------------------------------------------------------------------------------------
# the product's generated case table against the CLASSIC published table (oracle/mc_classic.py: what PyMCubes implements,
# lib/common/seg3d_lossless.py:592) - the triangulation stays unpinned (no PyMCubes here), but the gap is a LIST
# ---------------------------------------------------------------------------------------------
def _product_mesher(vol, level):
    """icon_export_mesh on one cube: export_mesh drops the first plane of every axis and returns (x, y, z) = array axes (2, 1, 0)"""
    P = np.zeros(tuple(s + 1 for s in vol.shape), np.float32)
    P[1:, 1:, 1:] = vol
    v, f = export_mesh_numpy(P, level)
    return v.numpy()[:, ::-1].copy(), f.numpy()


def test_classic_table_is_mechanically_valid():
    """the table is written out from the published one - so it is checked, not trusted: triangles only on cut edges, every cut
    edge used, closed fans, face segments that depend on the face's corner bits only (neighbouring cubes agree); and the classic
    algorithm on random volumes gives a closed, consistently oriented surface whose vertex set is every edge crossing"""
    from oracle import mc_classic
    r = mc_classic.validate_table()
    assert r["ok"], r["problems"][:5]
    assert r["separates_set_corners"]                       # on an ambiguous face the SET corners are cut off from each other
    rng = np.random.RandomState(5)
    for n, level in ((9, 0.5), (12, 0.37)):
        vol = rng.rand(n, n, n).astype(np.float32)
        for below in (True, False):
            v, f = mc_classic.marching_cubes(vol, level, set_below=below)
            P = np.zeros((n + 1,) * 3, np.float32)
            P[1:, 1:, 1:] = vol
            assert mc_check.same_point_set(v[:, ::-1], mc_check.edge_crossings(P, level))
            t = mc_check.topology(v, f, n)
            assert t["one_cube"] and t["used_all"] and t["oriented"] and t["closed"], (n, below, t)


def _cyc(t):
    t = list(t); k = t.index(min(t)); return tuple(t[k:] + t[:k])


def test_product_table_is_the_classic_table():
    """Round 5: the product's case table IS the published classic table, read as PyMCubes reads a numpy array (axes (0,1,2) =
    the cube's (x,y,z), bit set = corner at or below the level) and wound as the reference leaves it (faces[:, [0, 2, 1]],
    lib/common/seg3d_lossless.py:594): per cube configuration the same triangles, the same winding - all 256."""
    from oracle import mc_classic
    prod = mc_classic.case_tris_from_mesher(_product_mesher, set_is_inside=False)
    r = mc_classic.compare_tables(mc_classic.TRI_TABLE, prod)
    assert len(r["same"]) == 256, {k: v[:8] for k, v in r.items() if k != "same"}
    for c in range(256):
        assert {_cyc(t[[0, 2, 1]]) for t in mc_classic.TRI_TABLE[c]} == {_cyc(t) for t in prod[c]}, c


def test_generated_table_vs_classic_table():
    """The table of rounds 1-4 (ICON_AMD_MC_TABLE=generated: per-face marching squares, the INSIDE corners of an ambiguous face
    cut off from each other, ear-clipped loops) against the classic one, per configuration - the list DESIGN.md section 4.5
    quotes.  Under the reading "bit set = corner ABOVE the level": the same surface loops in all 256 configurations, 107 identical
    triangle sets, other diagonals in 149.  Under "bit set = corner BELOW the level" (Bourke's text; PyMCubes and kaolin as
    recalled): the 120 configurations with an ambiguous face are resolved the other way round, 82 more differ in diagonals."""
    code = ("import sys, json, numpy as np\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from oracle import mc_classic\n"
            "from test_mesh_tools import _product_mesher\n"
            "out = {}\n"
            "for name, inside in (('above', True), ('below', False)):\n"
            "    r = mc_classic.compare_tables(mc_classic.TRI_TABLE, mc_classic.case_tris_from_mesher(_product_mesher, set_is_inside=inside))\n"
            "    out[name] = r\n"
            "print('RESULT' + json.dumps(out))\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, ICON_AMD_MC_TABLE="generated")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1][6:])
    above, below = out["above"], out["below"]
    assert above["topology"] == [] and len(above["same"]) == 107 and len(above["triangulation"]) == 149
    assert len(below["topology"]) == 120 and len(below["same"]) == 54 and len(below["triangulation"]) == 82
    from oracle import mc_classic
    amb = [c for c in range(256) if any([(c >> m) & 1 for m in f] in ([1, 0, 1, 0], [0, 1, 0, 1]) for f in mc_classic.FACES.tolist())]
    assert sorted(below["topology"]) == amb          # they part exactly where a face is ambiguous


def test_classic_marching_cubes_on_the_body_volume():
    """the reference's own dense 33^3 volume: the classic algorithm (oracle/mc_classic.py, read as PyMCubes reads the array) and the
    product give the same vertex set AND the same triangles - compared as sets of vertex-position triples, all of them, the 28
    cells of this volume with an ambiguous face (of the 629 the surface passes through) included"""
    from oracle import mc_classic
    body = golden("seg3d_body_dense33.npz")["occ"]
    pv, pf = export_mesh_numpy(body, 0.5)
    pv = pv.numpy()[:, ::-1].astype(np.float64)                 # -> array-index order of the cropped volume
    s = body[1:, 1:, 1:] > 0.5
    idx = np.zeros(tuple(n - 1 for n in s.shape), np.int64)
    n0, n1, n2 = s.shape
    for m, (dx, dy, dz) in enumerate(mc_classic.CORNERS):
        idx |= s[dx:n0 - 1 + dx, dy:n1 - 1 + dy, dz:n2 - 1 + dz].astype(np.int64) << m
    amb = [c for c in range(256) if any([(c >> m) & 1 for m in f] in ([1, 0, 1, 0], [0, 1, 0, 1]) for f in mc_classic.FACES.tolist())]
    assert int(np.isin(idx, amb).sum()) == 28 and int(((idx > 0) & (idx < 255)).sum()) == 629
    cv, cf = mc_classic.marching_cubes(body[1:, 1:, 1:], 0.5, set_below=True)
    assert mc_check.same_point_set(cv, pv) and len(cf) == len(pf)
    key = lambda v, f: {tuple(sorted(map(tuple, np.round(v[t] * 4096).astype(np.int64).tolist()))) for t in f}
    assert key(cv, cf) == key(pv, pf.numpy())
    # the other reading of the case bit differs, and only by what those 28 cells can hold
    ov, of = mc_classic.marching_cubes(body[1:, 1:, 1:], 0.5, set_below=False)
    assert mc_check.same_point_set(ov, pv) and key(ov, of) != key(pv, pf.numpy()) and abs(len(of) - len(pf)) <= 2 * 28


def test_real_package_harness_absent_packages_and_difference_classes():
    """tools/parity_real_packages.py: in this image none of kaolin / PyMCubes / trimesh / pytorch3d / voxelize_cuda imports -
    every section says ABSENT and the command exits 0 (no device needed for that).  Its mesh comparison names the class of a
    difference: order only, constant offset, winding, missing triangles."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_real_packages.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    present = [m for m in ("kaolin", "mcubes", "trimesh", "pytorch3d", "voxelize_cuda") if __import__("importlib").util.find_spec(m)]
    if not present:
        assert p.returncode == 0, p.stdout[-800:]
        assert p.stdout.count("[ABSENT]") == 5 and "0 PASS, 0 DIFF, 5 ABSENT" in p.stdout, p.stdout[-800:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from parity_real_packages import compare_meshes
    v, f = export_mesh_numpy(np.random.RandomState(3).rand(17, 17, 17).astype(np.float32), 0.5)
    v, f = v.numpy(), f.numpy()
    same = compare_meshes(v, f, v, f)
    assert same["same_vertex_set"] and same["same_vertex_order"] and same["same_face_set"] and same["same_face_order"] and same["offset"] is None
    perm = np.random.RandomState(0).permutation(len(v))
    c = compare_meshes(v, f, v[perm], np.argsort(perm)[f][::-1])
    assert c["same_vertex_set"] and c["same_face_set"] and not c["same_vertex_order"] and not c["same_face_order"]
    c = compare_meshes(v, f, v + np.array([1.0, 0.0, 0.5]), f)
    assert c["offset"] == (1.0, 0.0, 0.5) and c["same_face_set"]
    c = compare_meshes(v, f, v, f[:, [0, 2, 1]])
    assert not c["same_face_set"] and c["flipped"] >= len(f) - 2 and c["only_ours"] == 0 and c["only_theirs"] == 0
    c = compare_meshes(v, f, v, f[:-7])
    assert not c["same_face_set"] and 5 <= c["only_ours"] <= 7 and c["only_theirs"] == 0
