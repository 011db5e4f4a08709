"""The per-image mesh preparation ON THE DEVICE (icon_amd/csrc/mesh_device.hip) against its checker, the host builder
(icon_amd/csrc/mesh_build.cpp, pure host code behind icon_debug_host_mesh_build): both must emit the same arena - vertex
normals, BVH nodes, leaf records, slot-ordered triangle records, inverse permutation, ray bins - BYTE FOR BYTE, and the
queries on a device-built mesh must equal the oracle as before.  Reference being replaced: the per-call prologue of
cal_sdf_batch, lib/dataset/mesh_util.py:367-372 (on the device there too)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from common import assets, orc, synth

pytestmark = pytest.mark.gpu

SECTIONS = ["dyn", "vnormals", "nodes", "leaves", "tris", "attr", "slot2face", "face2slot", "bin_start", "bin_slots"]


def dev():
    return torch.device("cuda:0")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def layout(V, F):
    from icon_amd import _lib
    lay = (C.c_int64 * 12)()
    _lib.check(_lib.lib().icon_debug_mesh_layout(C.c_int64(V), C.c_int64(F), lay))
    return list(lay)


def host_arena(v, f, cm, vs):
    from icon_amd import _lib
    lay = layout(len(v), len(f))
    ar = np.zeros(lay[11], np.uint8)
    _lib.check(_lib.lib().icon_debug_host_mesh_build(_lib.ptr(v), C.c_int64(len(v)), _lib.ptr(f), C.c_int64(len(f)), _lib.ptr(cm), _lib.ptr(vs),
                                                     _lib.ptr(ar), C.c_int64(len(ar))), "icon_debug_host_mesh_build")
    return lay, ar


class ZeroedMesh:
    """a device-built mesh whose arena was zeroed first (what the build does not write stays 0, as in the host arena)"""

    def __init__(self, v, f, cm, vs):
        from icon_amd import _lib
        from icon_amd.engine import _stream
        self.keep = [T(v), T(f), T(cm), T(vs)]
        self.lay = layout(len(v), len(f))
        self.arena = torch.zeros(self.lay[11], dtype=torch.uint8, device=dev())
        self.h = C.c_void_p(0)
        _lib.check(_lib.lib().icon_mesh_create_arena(_lib.ptr(self.keep[0]), C.c_int64(len(v)), _lib.ptr(self.keep[1]), C.c_int64(len(f)),
                                                     _lib.ptr(self.keep[2]), _lib.ptr(self.keep[3]), _lib.ptr(self.arena),
                                                     C.c_int64(self.lay[11]), _stream(), C.byref(self.h)), "icon_mesh_create_arena")
        bits = C.c_int(0)
        _lib.check(_lib.lib().icon_mesh_status(self.h, C.c_int(1), C.byref(bits)), "icon_mesh_status")
        self.bits = bits.value

    def close(self):
        from icon_amd import _lib
        _lib.lib().icon_mesh_destroy(self.h)


def mesh_arrays(name):
    if name in ("ico", "body"):
        a = assets(name)
        v, f = a.smpl_verts[0], a.smpl_faces[0]
        cm, vs = a.smpl_cmap[0], a.smpl_vis[0].reshape(-1)
    else:
        if name == "sphere6":                      # 81,920 faces: more slots than 15 bits, five multi-workgroup levels
            v, f = synth.icosphere(6, radius=0.62, center=(0.03, -0.05, 0.02))
            v = v * np.array([0.7, 1.25, 0.45])
        elif name == "tiny":                       # a tetrahedron: the root is a leaf
            v = np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5]], np.float64) - 0.1
            f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])
        elif name == "dup":                        # 3,000 copies of one triangle + a sphere: centroid extents of 0, positional splits
            v0, f0 = synth.icosphere(2, radius=0.3)
            f = np.concatenate([np.tile(f0[:1], (3000, 1)), f0])
            v = v0
        elif name == "line":                       # 2,400 slivers along a line: SAH peels them off one bin at a time (deep tree)
            t = np.geomspace(1e-6, 0.9, 2401)
            v = np.stack([np.concatenate([t, t]), np.concatenate([np.zeros_like(t), np.full_like(t, 1e-3)]), np.zeros(2 * len(t))], 1)
            i = np.arange(2400)
            f = np.stack([i, i + 1, i + 2401], 1)
        v, f = np.asarray(v, np.float32), np.asarray(f, np.int64)
        vs, cm = synth.make_vis_cmap(v, f)
        cm, vs = np.asarray(cm, np.float32).reshape(-1, 3), np.asarray(vs, np.float32).reshape(-1)
    return (np.ascontiguousarray(v, np.float32), np.ascontiguousarray(f, np.int64), np.ascontiguousarray(cm, np.float32),
            np.ascontiguousarray(vs, np.float32))


@pytest.mark.parametrize("name", ["ico", "body", "sphere6", "tiny", "dup", "line"])
def test_device_build_equals_host_build(name):
    v, f, cm, vs = mesh_arrays(name)
    lay, host = host_arena(v, f, cm, vs)
    m = ZeroedMesh(v, f, cm, vs)
    try:
        assert m.bits & ~4 == 0, f"status bits {m.bits}"
        devar = m.arena.cpu().numpy()
        for k, sec in enumerate(SECTIONS):
            a, b = lay[k], lay[k + 1] if sec != "dyn" else lay[k] + 128
            if sec == "dyn":                    # root, bin grid, box: bytes [0, 60); then the statistics
                hd, dd = host[a:b].view(np.int32), devar[a:b].view(np.int32)
                assert np.array_equal(hd[:15], dd[:15]), (name, hd[:15], dd[:15])
                assert np.array_equal(hd[15:20], dd[15:20]), (name, "stats", hd[15:20], dd[15:20])
                continue
            same = np.array_equal(host[a:b], devar[a:b])
            if not same:
                bad = np.nonzero(host[a:b] != devar[a:b])[0]
                dump = os.environ.get("ICON_AMD_DUMP_DIR")
                if dump:                        # both arenas for a post-mortem off the GPU box
                    os.makedirs(dump, exist_ok=True)
                    np.savez_compressed(os.path.join(dump, f"arena_{name}.npz"), host=host[: lay[10]], dev=devar[: lay[10]], lay=np.array(lay))
                raise AssertionError(f"{name}: section {sec} differs in {len(bad)} bytes, first at +{bad[0]} "
                                     f"(record {bad[0] // {'nodes': 64, 'leaves': 384, 'tris': 48, 'attr': 96}.get(sec, 4)}); dyn host {host[lay[0]:lay[0] + 84].view(np.int32)[[0, 15, 16, 17, 18, 19, 20]]} "
                                     f"dev {devar[lay[0]:lay[0] + 84].view(np.int32)[[0, 15, 16, 17, 18, 19, 20]]}")
    finally:
        m.close()


@pytest.mark.parametrize("name", ["tiny", "dup", "line", "body"])
def test_queries_on_device_built_mesh_vs_oracle(name):
    from icon_amd.engine import MeshHandle
    v, f, cm, vs = mesh_arrays(name)
    h = MeshHandle(T(v), T(f), T(cm), T(vs))
    rs = np.random.RandomState(4)
    lo, hi = v.min(0), v.max(0)
    pts = (lo + (hi - lo) * rs.rand(4000, 3) * 1.4 - 0.2 * (hi - lo)).astype(np.float32)
    g = {k: t.cpu().numpy() for k, t in h.sdf_query(T(pts)).items()}
    b = {k: t.cpu().numpy() for k, t in h.sdf_query(T(pts), search="brute").items()}
    d2, idx = orc.nearest_brute(v, f, pts)
    assert np.array_equal(g["face"], idx), f"{(g['face'] != idx).sum()} nearest-face mismatches"
    assert np.array_equal(b["face"], idx)
    assert np.array_equal(g["sdf"].view(np.uint32), b["sdf"].view(np.uint32))
    big = T(np.concatenate([pts] * 30))         # > 98,304 points: the packet traversal over the Morton order
    gb = h.sdf_query(big)["face"].cpu().numpy()
    assert np.array_equal(gb[: len(pts)], idx)


def test_full_query_on_a_mesh_whose_ray_bins_overflowed():
    """3,000 copies of one triangle: the ray-bin lists do not fit their buffer (kMeshBinOverflow), the inside test of EVERY
    kernel falls back to the brute-force parity count - the 4-lane sign kernel of small calls, the row crossings and the
    lattice sign kernel of slabs, the native schedule.  Occupancy vs the oracle (points and a 33^3 lattice), and the schedule
    against the host-driven one."""
    from types import SimpleNamespace
    from icon_amd.engine import IconQueryEngine, query_func
    from icon_amd.recon import AdaptiveReconEngine
    a = assets("body")
    v, f, cm, vs = mesh_arrays("dup")
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
    eng.set_mesh(T(v[None]), T(f[None]), T(cm[None]), T(vs[None, :, None]))
    eng.set_regressor({k: torch.from_numpy(x) for k, x in a.state_dict.items()})
    assert eng._mesh_handle().stats()["bin_entries"] == 0       # kMeshBinOverflow: no lists at all
    omlp = orc.Mlp(a.state_dict)
    rs = np.random.RandomState(6)
    pts = (rs.rand(5000, 3) * 0.9 - 0.45).astype(np.float32)
    occ = eng.query([T(a.features)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
    ref, _ = orc.query_icon(v, f, cm, vs, a.features, omlp, pts, sdf_clip=a.sdf_clip)
    assert np.abs(occ - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    from icon_amd import synth as S
    vol = eng.eval_slab(T(a.features), 33, 0, 33).cpu().numpy().ravel()
    ref33, _ = orc.query_icon(v, f, cm, vs, a.features, omlp, S.lattice_points(33), sdf_clip=a.sdf_clip)
    assert np.abs(vol - ref33).max() <= 1e-4 * max(1.0, float(np.abs(ref33).max()))
    kw = dict(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=[17, 33, 65], align_corners=True, faster=True)
    call = dict(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(a.features)], proj_matrix=None)
    nat, host = AdaptiveReconEngine(**kw).to(dev()), AdaptiveReconEngine(**kw).to(dev())
    host.native = False
    v1, v2 = nat(**call), host(**call)
    if v1 is None or v2 is None:
        assert v1 is None and v2 is None
    else:
        assert nat.last_stats["queries"] == host.last_stats["queries"] and (v1 - v2).abs().max().item() <= 1e-6


def test_bad_input_is_reported_not_faulted():
    """a face naming a missing vertex / a NaN coordinate: the build makes them harmless, the status says what was wrong
    (validate=True raises as the host build of round 3 did; the engine's lazy check raises one call late)"""
    from icon_amd.engine import IconAmdError, IconQueryEngine, MeshHandle
    a = assets("body")
    v, f, cm, vs = mesh_arrays("body")
    fb = f.copy(); fb[100, 1] = len(v) + 5
    with pytest.raises(IconAmdError, match="face index out of range"):
        MeshHandle(T(v), T(fb), T(cm), T(vs))
    vb = v.copy(); vb[17, 1] = np.nan
    with pytest.raises(IconAmdError, match="non-finite"):
        MeshHandle(T(vb), T(f), T(cm), T(vs))
    torch.cuda.synchronize()
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip)
    eng.set_mesh(T(vb[None]), T(f[None]), T(cm[None]), T(vs[None, :, None]))
    eng.set_regressor({k: torch.from_numpy(x) for k, x in a.state_dict.items()})
    feat = T(a.features)
    occ = eng.eval_slab(feat, 17, 0, 17)         # runs on the sanitised mesh: no fault
    torch.cuda.synchronize()
    assert torch.isfinite(occ).all()
    with pytest.raises(IconAmdError, match="non-finite"):
        eng.eval_slab(feat, 17, 0, 17)           # by now the build has reported


def test_mesh_create_does_not_wait_for_the_device():
    """icon_mesh_create_arena returns while a long kernel ahead of it on the stream is still running"""
    import time
    from icon_amd.engine import MeshHandle
    v, f, cm, vs = mesh_arrays("body")
    tv, tf, tc, ts = T(v), T(f), T(cm), T(vs)
    MeshHandle(tv, tf, tc, ts)                   # warm: module load, pinned pool, allocator cache
    x = torch.randn(8192, 8192, device=dev())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        x = x @ x * 1e-2                         # ~50 ms of queued work
    t1 = time.perf_counter()
    h = MeshHandle(tv, tf, tc, ts, validate=False)
    t2 = time.perf_counter()
    assert h.status() is None                    # not built yet, and asking did not block
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    assert h.status() == 0
    assert t3 - t2 > 5 * (t2 - t1), f"enqueue {1e3 * (t2 - t1):.2f} ms vs drain {1e3 * (t3 - t2):.2f} ms: the create call waited"
    assert t2 - t1 < 2e-3, f"icon_mesh_create_arena took {1e3 * (t2 - t1):.2f} ms of host time"
