"""CPU: the three third-party leaves of the oracle (kaolin point_to_mesh_distance / check_sign,
pytorch3d vertex normals) have no golden vectors upstream (PARITY UNPINNED), so
they are pinned mathematically here, against independent float64 numpy implementations that
share no code with oracle/icon_oracle.c."""
import numpy as np
import pytest

from common import assets, orc
from icon_amd import synth


def closest_point_f64(p, tri):
    """independent formulation: project onto the plane, fall back to the three segments"""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]            # [F,3]
    n = np.cross(b - a, c - a)
    nn = (n * n).sum(1)
    best = np.full(len(p), np.inf)
    arg = np.zeros(len(p), np.int64)
    for f in range(len(tri)):
        d = np.full(len(p), np.inf)
        if nn[f] > 0:
            t = ((p - a[f]) @ n[f]) / nn[f]
            q = p - t[:, None] * n[f]
            # inside test by sub-triangle orientation
            s0 = np.cross(b[f] - a[f], q - a[f]) @ n[f]
            s1 = np.cross(c[f] - b[f], q - b[f]) @ n[f]
            s2 = np.cross(a[f] - c[f], q - c[f]) @ n[f]
            inside = (s0 >= 0) & (s1 >= 0) & (s2 >= 0)
            d = np.where(inside, t * t * nn[f], np.inf)
        for u, v in ((a[f], b[f]), (b[f], c[f]), (c[f], a[f])):
            e = v - u
            tt = np.clip(((p - u) @ e) / max(e @ e, 1e-300), 0, 1)
            r = p - (u + tt[:, None] * e)
            d = np.minimum(d, (r * r).sum(1))
        upd = d < best
        best[upd], arg[upd] = d[upd], f
    return best, arg


@pytest.mark.parametrize("mesh", ["ico", "ico3"])
def test_nearest_triangle_distance(mesh):
    a = assets(mesh)
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    pts = synth.stratified_points(v, f, 600)
    d2, idx = orc.nearest_brute(v, f, pts)
    ref, arg = closest_point_f64(pts.astype(np.float64), v.astype(np.float64)[f])
    assert np.abs(np.sqrt(d2) - np.sqrt(ref)).max() <= 2e-6
    # the chosen face attains the minimum (ties may pick another face at equal distance)
    chosen, _ = closest_point_f64(pts.astype(np.float64), v.astype(np.float64)[f][idx][:, None].reshape(-1, 3, 3)[:1]) if False else (None, None)
    tri = v.astype(np.float64)[f]
    for i in range(0, len(pts), 7):
        di, _ = closest_point_f64(pts[i:i + 1].astype(np.float64), tri[idx[i]:idx[i] + 1])
        assert abs(np.sqrt(di[0]) - np.sqrt(ref[i])) <= 2e-6


def test_tie_rule_lowest_face_index():
    """a point straight above a shared vertex is equidistant from every triangle of the fan"""
    a = assets("ico")
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    vn = synth.vertex_normals_f64(v, f)
    for vid in (0, 5, 17, 100):
        p = (v[vid].astype(np.float64) + 0.2 * vn[vid]).astype(np.float32)[None]
        d2, idx = orc.nearest_brute(v, f, p)
        fan = np.where((f == vid).any(1))[0]
        per = np.array([orc.point_tri_dist2(p[0], v[f[k, 0]], v[f[k, 1]], v[f[k, 2]]) for k in fan], np.float32)
        assert idx[0] == fan[per == per.min()].min()
        assert d2[0] == per.min()


def _inside_analytic(points):
    """the icosphere asset is an affinely squashed + rotated sphere: exact inside test in its own
    frame (with a margin for the polyhedral approximation)"""
    rot = synth._rotation(np.random.RandomState(7), 11.0)
    q = (points.astype(np.float64) - np.array([0.03, -0.02, 0.01])) @ rot      # inverse rotation
    r = np.linalg.norm(q / (0.6 * np.array([0.7, 1.2, 0.5])), axis=1)
    return r


def test_check_sign_against_analytic_shape():
    a = assets("ico3")
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    rng = np.random.RandomState(2)
    pts = rng.uniform(-1, 1, (4000, 3)).astype(np.float32)
    r = _inside_analytic(pts)
    ins = orc.check_sign(v, f, pts)
    sure_in, sure_out = r < 0.95, r > 1.0          # level-3 icosphere is inscribed: radius >= 0.96 R
    assert ins[sure_in].all() and not ins[sure_out].any()


def test_check_sign_watertight_on_degenerate_rays():
    """rays through mesh vertices and along edges (the cases a naive Moeller-Trumbore parity gets
    wrong): the canonical-edge rule must count each crossing exactly once"""
    a = assets("ico")
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    pts = []
    for vid in range(0, len(v), 3):
        pts.append(v[vid] + np.float32([-3.0, 0, 0]))          # ray passes exactly through vertex vid
        pts.append(v[vid] + np.float32([-1e-3, 0, 0]))
    for k in range(0, len(f), 5):
        m = 0.5 * (v[f[k, 0]] + v[f[k, 1]])                    # ... and through an edge midpoint
        pts.append(m + np.float32([-3.0, 0, 0]))
    pts = np.array(pts, np.float32)
    ins = orc.check_sign(v, f, pts)
    r = _inside_analytic(pts)
    far = np.abs(pts[:, 0]) > 1.5
    assert not ins[far].any()
    sure = (r < 0.9) | (r > 1.02)
    assert np.array_equal(ins[sure], (r < 0.9)[sure])


def test_vertex_normals():
    a = assets("body")
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    vn = orc.vertex_normals(v, f)
    ref = synth.vertex_normals_f64(v, f)
    assert np.abs(vn - ref).max() <= 5e-6
    assert np.abs(np.linalg.norm(vn, axis=1) - 1).max() <= 1e-6


def test_body_mesh_sign_matches_star_shape():
    """the synthetic body is star-shaped about its centre: inside <=> |p - c| < r(direction)"""
    a = assets("body")
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    rng = np.random.RandomState(4)
    pts = rng.uniform(-1, 1, (1500, 3)).astype(np.float32)
    ins = orc.check_sign(v, f, pts)
    # undo the small rotation / shift of generate_body_mesh
    R = synth._rotation(np.random.RandomState(synth.SEED), 4.0)
    q = (pts.astype(np.float64) - np.array([0.00317, -0.00211, 0.00473])) @ R - synth._CENTER
    rad = np.linalg.norm(q, axis=1)
    lim = synth._radial(q / rad[:, None])
    sure = np.abs(rad / lim - 1) > 0.08
    assert np.array_equal(ins[sure], (rad < lim)[sure])
    assert ins.sum() > 5


@pytest.mark.parametrize("mesh", ["body", "ico"])
def test_accelerated_leaves_equal_the_linear_scans(mesh):
    """oracle/icon_accel.c (BVH nearest + binned ray parity) against the definitions in icon_oracle.c: squared
    distance bit for bit, the same face index (lowest on float32 ties), the same inside flag - on
    near-surface / far / boundary points, lattice points, mesh vertices and edge midpoints (exact ties,
    rays through vertices and edges)."""
    a = synth.make_assets(mesh)
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    pts = np.concatenate([synth.stratified_points(v, f, 3000), synth.lattice_points(17), v[:400],
                          ((v[f[:400, 0]] + v[f[:400, 1]]) / 2).astype(np.float32)])
    d2b, ib = orc.nearest_brute(v, f, pts)
    sb = orc.check_sign(v, f, pts)
    acc = orc.Accel(v, f)
    d2a, ia = acc.nearest(pts)
    sa = acc.check_sign(pts)
    assert np.array_equal(d2a.view(np.int32), d2b.view(np.int32))
    assert np.array_equal(ia, ib)
    assert np.array_equal(sa, sb)


def test_cal_sdf_accel_switch_is_invisible():
    a = synth.make_assets("body")
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    pts = synth.stratified_points(v, f, 1500, seed=3)
    try:
        orc.set_accel(False)
        slow = orc.cal_sdf(v, f, a.smpl_cmap[0], a.smpl_vis[0], pts)
    finally:
        orc.set_accel(True)
    fast = orc.cal_sdf(v, f, a.smpl_cmap[0], a.smpl_vis[0], pts)
    for k in slow:
        assert np.array_equal(slow[k], fast[k]), k


def test_query_icon_subset_equals_full_call():
    """the subset form (bench.py's live parity sample) returns exactly the rows of the full call"""
    a = synth.make_assets("body")
    pts = synth.lattice_points(17)
    mlp = orc.Mlp(a.state_dict)
    occ, X = orc.query_icon(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], a.features, mlp, pts)
    sub = np.random.RandomState(0).choice(len(pts), 700, replace=False)
    occ_s, X_s = orc.query_icon_subset(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], a.features, mlp,
                                       pts, sub)
    assert np.array_equal(X_s, X[sub]) and np.array_equal(occ_s, occ[sub])


def _seg_dist2_f64(p, a, b):
    p, a, b = (np.asarray(x, np.float64) for x in (p, a, b))
    e = b - a
    l2 = e @ e
    t = 0.0 if l2 == 0 else min(max((p - a) @ e / l2, 0.0), 1.0)
    d = p - (a + t * e)
    return d @ d


def test_degenerate_triangles_are_their_edge_segments():
    """zero-area triangles (collinear corners in any order, a repeated vertex, three equal vertices): the distance is
    the distance to the union of the three edge segments - pinned against float64, not merely HIP == checker"""
    rng = np.random.RandomState(3)
    a = np.array([0.1, -0.2, 0.05], np.float32)
    d = np.array([0.5, 0.25, -0.125], np.float32)                       # exactly representable multiples: exactly collinear
    cases = [(a, a + d, a + 2 * d), (a, a + 2 * d, a + d), (a + d, a, a + 2 * d),        # the far corner first / middle / last
             (a, a + d, a + d), (a + d, a, a + d), (a + d, a + d, a),                    # repeated vertex
             (a, a, a)]
    for A, B, C in cases:
        for _ in range(200):
            p = (a + rng.normal(0, 0.6, 3)).astype(np.float32)
            want = min(_seg_dist2_f64(p, A, B), _seg_dist2_f64(p, A, C), _seg_dist2_f64(p, B, C))
            got = orc.point_tri_dist2(p, A, B, C)
            assert abs(got - want) <= 1e-6 * max(want, 1e-3), (A, B, C, p, got, want)
