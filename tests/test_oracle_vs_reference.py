"""CPU, build container only: the oracle against the reference's own Python imported verbatim
from /root/reference (skipped on machines without the reference tree, e.g. the GPU box - the
committed fixtures in tests/golden/ carry the same information there)."""
import os

import numpy as np
import pytest
import torch

from common import assets, oracle_query, orc
from icon_amd import synth
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    return ref_loader.load()


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_query_func_fresh_points(ref):
    a = assets("ico")
    netG, cfg = ref_loader.build_netG(a)
    pts = synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], 3000, seed=99)
    with torch.no_grad():
        out = ref.query_func(cfg, netG, [T(a.features)], T(pts)[None])[0, 0].numpy()
    occ, _ = oracle_query(a, pts)
    assert np.abs(occ - out).max() <= 2e-6


def test_projection_arithmetic_matches_aten(ref):
    """orc_project == torch.baddbmm on CPU, bit for bit (the spec's fma-chain-then-add)"""
    rng = np.random.RandomState(3)
    A = np.eye(4, dtype=np.float32)
    A[:3, :3] += rng.normal(0, 0.1, (3, 3)).astype(np.float32)
    A[:3, 3] = rng.normal(0, 0.05, 3).astype(np.float32)
    p = rng.uniform(-1, 1, (5000, 3)).astype(np.float32)
    t = ref.orthogonal(T(p.T.copy())[None], T(A)[None])[0].T.numpy()
    import ctypes as C
    out = np.empty_like(p)
    cal = np.ascontiguousarray(A[:3])
    orc.lib().orc_project(cal.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), C.c_int64(len(p)),
                          out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, t)


def test_adaptive_seg3d_vs_dense(ref):
    """the reference's coarse-to-fine loop evaluates the same field: on the points it actually
    queries (not interpolates) at the last queried level it agrees with the dense evaluation up to
    the batch-dependent cmap tiling; globally the two volumes describe the same surface"""
    a = assets("body")
    netG, cfg = ref_loader.build_netG(a)
    with torch.no_grad():
        eng = ref.Seg3dLossless(query_func=ref.query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                resolutions=[17, 33], align_corners=True, balance_value=0.5, faster=True)
        vol = eng(opt=cfg, netG=netG, features=[T(a.features)], proj_matrix=None).numpy()
    dense, _ = oracle_query(a, synth.lattice_points(33), cmap_local=True)
    dense = dense.reshape(33, 33, 33)
    agree = ((vol > 0.5) == (dense > 0.5)).mean()
    assert agree > 0.995


def test_chamfer_dense_vs_adaptive_cpu(ref):
    """mesh of the dense field (oracle) vs mesh of the reference's adaptive field at 33^3: the number
    DESIGN.md quotes for 'mesh Chamfer vs ref' at CPU-test size"""
    from common import chamfer
    from icon_amd.recon import export_mesh_numpy
    a = assets("body")
    netG, cfg = ref_loader.build_netG(a)
    with torch.no_grad():
        eng = ref.Seg3dLossless(query_func=ref.query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                resolutions=[17, 33], align_corners=True, balance_value=0.5, faster=True)
        adaptive = eng(opt=cfg, netG=netG, features=[T(a.features)], proj_matrix=None).numpy()
    dense, _ = oracle_query(a, synth.lattice_points(33))
    dense = dense.reshape(33, 33, 33)

    def mesh(vol):
        v, f = export_mesh_numpy(np.ascontiguousarray(vol, dtype=np.float32), 0.5)
        return ((v.numpy() - 16.0) / 16.0).astype(np.float32), f.numpy()

    (vd, fd), (va, fa) = mesh(dense), mesh(adaptive)
    c, _ = chamfer(vd, fd, va, fa, n=8000)
    assert c <= 3.2, c          # half a voxel: the voxel size at 33^3 is 6.25 on this x100 scale


def test_reference_transforms_branch_is_dead_code(ref):
    """lib/net/geometry.py:57-60: orthogonal(points, calibs, transforms) raises for the documented [B,2,3]
    layout and for a bare [2,3] matrix alike, so HGPIFuNet.query(transforms=...) has no behaviour to mirror;
    IconQueryEngine.query raises a clear error for it (icon_amd/engine.py)."""
    pts = torch.randn(1, 3, 8)
    calib = torch.eye(4)[None]
    for tr in (torch.tensor([[1.0, 0.0, 0.1], [0.0, 1.0, -0.1]]), torch.tensor([[[1.0, 0.0, 0.1], [0.0, 1.0, -0.1]]])):
        with pytest.raises(Exception):
            ref.orthogonal(pts, calib, tr)
    from common import ROOT
    src = open(os.path.join(ROOT, "icon_amd", "engine.py")).read()
    assert "transforms is not None" in src and "is not supported" in src
