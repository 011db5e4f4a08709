"""GPU tests of round 5: the per-call tile record of the 257^3-class search, the bounded hand-over of the shared walks
(torture, fault injection, error reporting through the C ABI), the mcube_res=512 path end to end (reference schedule
[33,65,129,257,513] -> device marching cubes -> clean_mesh), lazy mesh validation."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from common import assets, golden
from icon_amd import _lib, synth
from test_gpu_parity import T, dev, make_engine

pytestmark = pytest.mark.gpu


def set_option(key, value):
    _lib.check(_lib.lib().icon_debug_set_option(key.encode(), C.c_int(int(value))), "icon_debug_set_option")


@pytest.fixture()
def options():
    """test switches of the library are process-wide: whatever a test sets is put back"""
    touched = []

    def setter(key, value):
        touched.append(key)
        set_option(key, value)
    yield setter
    defaults = dict(lattice_fast=1, share_waves=-1, share_ring=0, share_lose_push=0, share_spin_log2=0)
    for k in touched:
        set_option(k, defaults[k])


@pytest.fixture(scope="module")
def body():
    return assets("body")


# ---------------------------------------------------------------------------------------------
# k_nearest<lattice>: the packet set-up read from the per-call record == the set-up every packet derives itself
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("res,z0,z1", [(129, 0, 129), (257, 0, 40), (257, 101, 157), (257, 230, 257), (161, 3, 160)])
def test_lattice_fast_setup_names_the_same_points(body, options, res, z0, z1):
    """LatticeFast (geom_device.h): trimmed region, tile counts and multiply-high reciprocals written once per call by the
    row-crossings kernel; a packet decodes its tile with three s_mul_hi_u32.  Same tiles, same rotation, same points: the
    slab's volume and its feature rows must not change by one bit (whole slabs, slabs inside the body, slabs that are
    mostly shell, a resolution whose tile counts are not powers of two)."""
    feat = T(body.features)
    eng = make_engine(body)
    options("lattice_fast", 0)
    want = eng.eval_slab(feat, res, z0, z1).clone()
    want_rows = eng._rows(feat, lattice=(res, z0, min(z1, z0 + 24))).clone()
    options("lattice_fast", 1)
    got = eng.eval_slab(feat, res, z0, z1)
    got_rows = eng._rows(feat, lattice=(res, z0, min(z1, z0 + 24)))
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert torch.equal(got_rows.view(torch.int32), want_rows.view(torch.int32))
    assert float(want.max()) > 0.5 or z1 - z0 < 60


def test_lattice_fast_setup_under_the_alternative_tie_rule(body, options):
    """the ALT instantiation of the kernel takes the record too"""
    feat = T(body.features)
    eng = make_engine(body)
    eng.tie_rule = ("highest", 1)
    options("lattice_fast", 0)
    want = eng.eval_slab(feat, 129, 0, 129).clone()
    options("lattice_fast", 1)
    assert torch.equal(eng.eval_slab(feat, 129, 0, 129).view(torch.int32), want.view(torch.int32))


# ---------------------------------------------------------------------------------------------
# shared walks: torture + fault injection
# ---------------------------------------------------------------------------------------------
def small_mesh(name):
    if name == "one":                               # a single triangle: the root is a leaf of one
        v = np.array([[-0.3, -0.2, 0.05], [0.4, -0.1, 0.0], [0.0, 0.5, -0.1]], np.float32)
        f = np.array([[0, 1, 2]], np.int64)
    elif name == "tetra":
        v = (np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5]], np.float32) - 0.1).astype(np.float32)
        f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int64)
    elif name == "ico20":
        v, f = synth.icosphere(0, radius=0.55)
    elif name == "open64":                          # the first 64 faces of an 80-face sphere: an open surface
        v, f = synth.icosphere(1, radius=0.5)
        f = f[:64]
    elif name == "fan":                             # 48 triangles around one vertex, every third one of zero area, a few repeated
        n = 48
        ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
        rim = np.stack([0.6 * np.cos(ang), 0.6 * np.sin(ang), 0.1 * np.sin(3 * ang)], 1)
        rim[2::3] = rim[1::3]                       # coincident rim vertices -> zero-area triangles
        v = np.concatenate([[[0.0, 0.0, 0.2]], rim]).astype(np.float32)
        i = np.arange(n)
        f = np.stack([np.zeros(n, np.int64), 1 + i, 1 + (i + 1) % n], 1)
        f = np.concatenate([f, f[:5]])
    v, f = np.ascontiguousarray(v, np.float32), np.ascontiguousarray(f, np.int64)
    vis, cmap = synth.make_vis_cmap(v, f)
    return v, f, np.asarray(cmap, np.float32).reshape(-1, 3), np.asarray(vis, np.float32).reshape(-1, 1)


def engine_for(v, f, cm, vs, a):
    from icon_amd.engine import IconQueryEngine
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
    eng.set_mesh(T(v)[None], T(f)[None], T(cm)[None], T(vs)[None])
    eng.set_regressor({k: torch.from_numpy(x) for k, x in a.state_dict.items()})
    return eng


def work_error(eng):
    """the workspace's verdict on everything launched so far: None, or the message of ICON_ERR_STATE (cleared by the read)"""
    from icon_amd.engine import IconAmdError
    torch.cuda.synchronize()
    try:
        eng._work().status()
    except IconAmdError as e:
        return str(e)
    return None


@pytest.mark.parametrize("name", ["one", "tetra", "ico20", "open64", "fan"])
@pytest.mark.parametrize("nw", [8, 16])
def test_shared_walk_torture_small_meshes(body, options, name, nw):
    """meshes of 1-69 triangles (root = leaf, trees of two or three levels, zero-area and repeated triangles), the walk of
    every packet forced through the 8- / 16-wave hand-over, with the production ring (64 slots: NO error may be reported and
    the rows must equal the point-mode rows bit for bit) and with rings of 2 and 4 slots - fewer slots than waves, so pushers
    really wait for poppers and may give up: the launch must END, and it must either be clean and exact or SAY so."""
    v, f, cm, vs = small_mesh(name)
    eng = engine_for(v, f, cm, vs, body)
    feat = T(body.features)
    res = 33
    pts = synth.lattice_points(res)
    options("share_waves", 1)
    want = eng._rows(feat, points=T(pts), calib12=np.eye(4, dtype=np.float32)[:3].copy()).cpu().numpy()
    assert work_error(eng) is None
    options("share_waves", nw)
    for ring, spin in ((0, 0), (4, 12), (2, 12)):
        options("share_ring", ring)
        options("share_spin_log2", spin)
        for rep in range(3):
            got = eng._rows(feat, lattice=(res, 0, res)).cpu().numpy()
            err = work_error(eng)
            if ring == 0:
                assert err is None, err
            if err is None:
                bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(1))[0]
                assert len(bad) == 0, f"{name} nw={nw} ring={ring} pass {rep}: {len(bad)} rows differ, first {bad[:5]}"
            else:
                assert "shared-walk search" in err and ring in (2, 4)


@pytest.mark.parametrize("mesh", ["body", "ico"])
@pytest.mark.parametrize("nw", [8, 16])
def test_shared_walk_tiny_ring_real_meshes(options, mesh, nw):
    """the real trees (4,500 nodes, walks of up to 2,000 visits) through a 2-slot ring with a short wait bound: terminates,
    and a launch that reports nothing is exact"""
    a = assets(mesh)
    eng = make_engine(a)
    feat = T(a.features)
    res = 33
    options("share_waves", 1)
    want = eng._rows(feat, lattice=(res, 0, res)).cpu().numpy()
    options("share_waves", nw)
    options("share_ring", 2)
    options("share_spin_log2", 10)
    clean = 0
    for rep in range(4):
        got = eng._rows(feat, lattice=(res, 0, res)).cpu().numpy()
        err = work_error(eng)
        if err is None:
            clean += 1
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    options("share_ring", 0)
    options("share_spin_log2", 0)
    got = eng._rows(feat, lattice=(res, 0, res)).cpu().numpy()
    assert work_error(eng) is None and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_lost_hand_over_is_reported_not_hung(body, options):
    """fault injection: the push with ticket 3 of every packet claims its ticket and announces it (avail + 1) but never
    stores the node - exactly what a lost LDS store or a protocol bug would look like.  Before round 5 the popper of that
    ticket span forever (GPU hang -> SIGKILL, no message).  Now: the launch ends within its wait bound, the workspace's error
    record names the wait, the C ABI returns ICON_ERR_STATE - from icon_work_status, from the NEXT call on the workspace, and
    from icon_adaptive_eval for the schedule it has just synchronised on - and the workspace works again afterwards."""
    from icon_amd.engine import IconAmdError
    feat = T(body.features)
    eng = make_engine(body)
    want = eng.eval_slab(feat, 33, 0, 33).clone()
    assert work_error(eng) is None
    options("share_spin_log2", 10)
    options("share_lose_push", 3)
    # (a) a coarse slab call: asynchronous - the verdict comes from icon_work_status after a synchronisation ...
    eng.eval_slab(feat, 33, 0, 33)
    err = work_error(eng)
    assert err is not None and "lost or abandoned push" in err and "code 1" in err and "ticket 3" in err, err
    assert work_error(eng) is None                  # the report clears the record
    # ... or from the next compute call on the workspace, without anybody asking
    eng.eval_slab(feat, 33, 0, 33)
    torch.cuda.synchronize()
    with pytest.raises(IconAmdError, match="shared-walk search"):
        eng.eval_slab(feat, 33, 0, 33)
    torch.cuda.synchronize()
    work_error(eng)                                 # (the refused call launched nothing; drop whatever the one before left)
    # (b) the native schedule synchronises for its counts: it reports its OWN searches
    with pytest.raises(IconAmdError, match="shared-walk search"):
        eng.adaptive_eval(feat, [33, 65, 129], 0.5)
    work_error(eng)
    # (c) switch the fault off: same workspace, exact results again
    options("share_lose_push", 0)
    options("share_spin_log2", 0)
    got = eng.eval_slab(feat, 33, 0, 33)
    assert work_error(eng) is None
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    vol, counts, any_pos = eng.adaptive_eval(feat, [33, 65, 129], 0.5)
    assert any_pos and counts[0] == 33 ** 3


# ---------------------------------------------------------------------------------------------
# lazy mesh validation: every mesh is checked by the end of ITS image
# ---------------------------------------------------------------------------------------------
def test_bad_mesh_is_reported_by_the_call_that_used_it(body):
    """the engine builds with validate=False and used to poll the status only on a LATER call with the same
    mesh - apps bind new SMPL tensors per image and make one adaptive_eval per mesh, so a face naming a missing vertex was
    never reported (the device build clamps it to vertex 0: a plausible, wrong volume).  Now adaptive_eval (it synchronises for
    its counts), reconEngine.forward (its None test) and the replacement of an unchecked mesh all poll."""
    from icon_amd.engine import IconAmdError, IconQueryEngine, query_func
    from icon_amd.recon import AdaptiveReconEngine, DenseReconEngine
    feat = T(body.features)
    bad_faces = body.smpl_faces.copy()
    bad_faces[0, 100, 1] = body.smpl_verts.shape[1] + 7

    def engine(faces):
        eng = IconQueryEngine(prior_type="icon", sdf_clip=body.sdf_clip)
        eng.set_mesh(T(body.smpl_verts), T(faces), T(body.smpl_cmap), T(body.smpl_vis))
        eng.set_regressor({k: torch.from_numpy(v) for k, v in body.state_dict.items()})
        return eng
    with pytest.raises(IconAmdError, match="face index out of range"):
        engine(bad_faces).adaptive_eval(feat, [33, 65], 0.5)
    eng = engine(bad_faces)
    rec = DenseReconEngine(query_func=query_func, resolutions=[33], align_corners=True, engine=eng).to(dev())
    with pytest.raises(IconAmdError, match="face index out of range"):
        rec(opt=SimpleNamespace(num_views=1), netG=eng, features=[feat], proj_matrix=None)
    # one asynchronous call on a bad mesh, then the next image's tensors: the replacement reports the mesh it drops
    eng = engine(bad_faces)
    eng.eval_slab(feat, 33, 0, 33)
    eng.set_mesh(T(body.smpl_verts), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))
    with pytest.raises(IconAmdError, match="PREVIOUS SMPL mesh"):
        eng.eval_slab(feat, 33, 0, 33)
    assert float(eng.eval_slab(feat, 33, 0, 33).max()) > 0.5       # the good mesh is bound and evaluated
    ad = AdaptiveReconEngine(faster=True, query_func=query_func, resolutions=[33, 65], align_corners=True, engine=eng).to(dev())
    assert ad(opt=SimpleNamespace(num_views=1), netG=eng, features=[feat], proj_matrix=None) is not None and ad.last_stats["native"] is True


# ---------------------------------------------------------------------------------------------
# mcube_res = 512, the SHIPPED default (configs/icon-filter.yaml:23): apps/ICON.py:62-72 builds the schedule
# [33, 65, 129, 257, 513]; lib/common/seg3d_lossless.py:587-596 then takes the PyMCubes branch on the CPU (final.shape[0] != 256);
# apps/ICON.py:755-756 cleans the mesh.  End to end on the device: native schedule -> device marching cubes -> icon_clean_mesh.
# ---------------------------------------------------------------------------------------------
RES513 = [33, 65, 129, 257, 513]


def recon513(eng, native=True):
    from icon_amd.engine import query_func
    from icon_amd.recon import AdaptiveReconEngine
    ad = AdaptiveReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=RES513,
                             align_corners=True, balance_value=0.5, faster=True).to(dev())
    ad.native = native
    return ad


@pytest.fixture(scope="module")
def vol513(body):
    eng = make_engine(body)
    ad = recon513(eng)
    vol = ad(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
    return vol, dict(ad.last_stats), eng


def test_adaptive_recon_513_matches_reference_schedule(vol513):
    """Seg3dLossless [33,65,129,257,513], faster=True, run verbatim on the synthetic subject by tools/make_golden.py (k): the ONE-
    CALL native schedule queries the same number of points at every level (35,937 / 20,052 / 62,703 / 142,857; the 513^3 level is
    interpolated only) and gives the same volume on the stored subsets: stride-8 sub-lattice, three mid planes, 60,000 random
    voxels, 60,000 voxels of the level-set band"""
    from test_gpu_parity import OCC_TOL
    g = golden("seg3d_body_adaptive_513.npz")
    vol, stats, _ = vol513
    assert vol.shape == (513, 513, 513)
    assert stats.get("native") is True, stats
    assert stats["queries"] == [int(q) for q in g["queries"]]
    flat = vol.reshape(-1)
    for name, got in (("sub8", vol[::8, ::8, ::8]), ("plane_z", vol[256]), ("plane_y", vol[:, 256]), ("plane_x", vol[:, :, 256]),
                      ("samples", flat[torch.from_numpy(g["idx"]).to(flat.device)]),
                      ("band_samples", flat[torch.from_numpy(g["band_idx"]).to(flat.device)])):
        d = np.abs(got.cpu().numpy() - g[name]).max()
        assert d <= OCC_TOL, (name, d)
    assert abs(int((vol > 0.5).sum()) - int(g["inside"])) <= 16          # voxels within 1e-4 of the level may flip
    assert abs(float(flat.double().sum()) - float(g["vol_sum"])) <= 1e-6 * 513 ** 3


def test_native_schedule_513_equals_host_driven_schedule(vol513, body):
    """the five-level schedule through the host-driven form (torch bookkeeping around HIP queries): same points per level, same
    volume <= 1e-6; and the native call again on its reused buffers: the same bits"""
    vol, stats, eng = vol513
    call = dict(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
    host = recon513(eng, native=False)
    v2 = host(**call)
    assert host.last_stats["native"] is False and host.last_stats["queries"] == stats["queries"]
    assert (vol - v2).abs().max().item() <= 1e-6
    del v2
    again = recon513(eng)(**call)
    assert torch.equal(again, vol)


@pytest.mark.parametrize("which", ["schedule", "dense"])
def test_device_marching_cubes_at_513(vol513, body, which):
    """device marching cubes on a 513^3 volume (the reference runs PyMCubes on the CPU there): == the host implementation as
    sets (vertices bit for bit, triangles), and for the schedule's volume the table-free invariants of oracle/mc_check.py:
    the vertex set is every lattice-edge crossing at its linear interpolation, one-cube triangles, closed and consistently
    oriented, outward normals"""
    from icon_amd.recon import export_mesh_device, export_mesh_numpy
    from oracle import mc_check
    from test_gpu_parity import _canonical_mesh
    vol, _, eng = vol513
    occ = vol if which == "schedule" else eng.eval_slab(T(body.features), 513, 0, 513)
    vd, fd = export_mesh_device(occ, 0.5)
    occ_h = occ.cpu().numpy()
    vh, fh = export_mesh_numpy(occ_h, 0.5)
    assert vd.shape == vh.shape and fd.shape == fh.shape and len(fh) > 400000
    vdn, fdn = vd.cpu().numpy(), fd.cpu().numpy()
    a, b = _canonical_mesh(vdn, fdn), _canonical_mesh(vh.numpy(), fh.numpy())
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if which == "schedule":
        assert mc_check.same_point_set(vdn.astype(np.float64), mc_check.edge_crossings(occ_h, 0.5))
        t = mc_check.topology(vdn, fdn, 512)
        assert t["one_cube"] and t["used_all"] and t["oriented"] and t["closed"] and t["watertight"], t
        assert t["signed_volume"] > 0 and t["euler"] % 2 == 0 and t["euler"] <= 2 * t["components"], t


def test_clean_mesh_at_513(vol513):
    """icon_clean_mesh on the 513^3 surface (580,000 faces) == trimesh's rules spelled out in plain Python
    (oracle/mc_check.py: largest_component_by_faces), vertex for vertex and face for face; what it drops is small"""
    from icon_amd.recon import clean_mesh, export_mesh_device
    from oracle.mc_check import largest_component_by_faces
    vol, _, _ = vol513
    v, f = export_mesh_device(vol, 0.5)
    cv, cf = clean_mesh(v, f)
    ev, ef = largest_component_by_faces(v.cpu().numpy(), f.cpu().numpy())
    assert np.array_equal(cv.cpu().numpy(), ev) and np.array_equal(cf.cpu().numpy(), ef)
    assert len(ef) > 0.95 * len(f)
