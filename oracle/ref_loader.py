"""Import the reference's OWN Python for the query path, verbatim, on CPU - build container only.

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box, so this module is
used (a) by tools/make_golden.py to generate the fixtures under tests/golden/, and (b) by the
``not gpu`` tests that pin oracle/icon_oracle.c against the reference where the tree is present
(they skip elsewhere).  Nothing is copied from the reference: its modules are imported from
where they lie, with ``sys.modules`` stubs for the third-party packages that are not installed
(recipe: SURVEY.md §8c).

The three third-party leaves the reference calls are bound to the oracle's CPU restatements:
    kaolin.metrics.trianglemesh.point_to_mesh_distance  -> oracle.nearest_brute
    kaolin.ops.mesh.check_sign                          -> oracle.check_sign
    pytorch3d.structures.Meshes.verts_normals_padded    -> oracle.vertex_normals
so everything ABOVE the leaves (barycentric interpolation, clipping, channel order, feature
select, grid_sample, MLP, in-cube mask, lattice mapping, the adaptive Seg3dLossless loop) is the
reference's own code.
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

REFERENCE_ROOT = os.environ.get("ICON_REFERENCE_ROOT", "/root/reference")

_loaded = None
ACCEL = [True]


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lib", "net"))


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so sub-imports resolve through sys.modules
    sys.modules[name] = m
    return m


def load():
    """Returns a namespace with the reference's HGPIFuNet, query_func, Seg3dLossless,
    cal_sdf_batch, feat_select, MLP, index, orthogonal."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True          # never drop __pycache__ into the reference tree
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    import numpy as np
    import torch

    if not hasattr(np, "long"):
        np.long = np.int64                  # seg3d_lossless.py:594 uses the removed alias

    here = os.path.dirname(os.path.abspath(__file__))
    if os.path.dirname(here) not in sys.path:
        sys.path.insert(0, os.path.dirname(here))
    from oracle import oracle as orc

    # ---- leaves -------------------------------------------------------------------
    # ACCEL[0]: answer the two O(N*F) leaves through oracle/icon_accel.c (BVH / ray bins; bit-identical
    # to the linear scans, tests/test_oracle_leaves.py) - what bench.py's timed "reference" leg uses
    def point_to_mesh_distance(points, triangles):
        # kaolin signature: points [B,N,3], face_vertices [B,F,3,3] -> (dist2 [B,N], idx [B,N], type)
        assert points.shape[0] == 1
        tri = triangles[0].detach().cpu().numpy().astype(np.float32)
        verts = tri.reshape(-1, 3)
        faces = np.arange(len(verts), dtype=np.int64).reshape(-1, 3)
        pts = points[0].detach().cpu().numpy()
        d2, idx = orc.Accel(verts, faces).nearest(pts) if ACCEL[0] else orc.nearest_brute(verts, faces, pts)
        return (torch.from_numpy(d2)[None], torch.from_numpy(idx)[None],
                torch.zeros(1, len(d2), dtype=torch.int32))

    def check_sign(verts, faces, points, hash_resolution=512):
        assert verts.shape[0] == 1 and faces.dim() == 2
        v, f, pts = verts[0].detach().cpu().numpy(), faces.detach().cpu().numpy(), points[0].detach().cpu().numpy()
        ins = orc.Accel(v, f).check_sign(pts) if ACCEL[0] else orc.check_sign(v, f, pts)
        return torch.from_numpy(ins)[None]

    class Meshes:
        def __init__(self, verts, faces, **kw):
            self._verts, self._faces = verts, faces

        def verts_normals_padded(self):
            out = [torch.from_numpy(orc.vertex_normals(v.detach().cpu().numpy(), f.detach().cpu().numpy()))
                   for v, f in zip(self._verts, self._faces)]
            return torch.stack(out).type_as(self._verts)

    # ---- stubs for packages that are not installed -----------------------------------
    ident = lambda x, *a, **k: x
    _stub("cv2")
    _stub("pymeshlab")
    _stub("torchvision"); _stub("torchvision.models")
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    _stub("trimesh")
    _stub("termcolor", colored=ident)
    _stub("pytorch3d"); _stub("pytorch3d.io"); _stub("pytorch3d.loss")
    _stub("pytorch3d.structures", Meshes=Meshes)
    _stub("pytorch3d.renderer"); _stub("pytorch3d.renderer.mesh", rasterize_meshes=None)
    _stub("kaolin"); _stub("kaolin.ops"); _stub("kaolin.ops.conversions", voxelgrids_to_trianglemeshes=None)
    _stub("kaolin.ops.mesh", check_sign=check_sign)
    _stub("kaolin.metrics"); _stub("kaolin.metrics.trianglemesh", point_to_mesh_distance=point_to_mesh_distance)
    _stub("pytorch_lightning", LightningModule=torch.nn.Module)
    _stub("voxelize_cuda")
    _stub("mcubes")
    _stub("rtree"); _stub("skimage"); _stub("skimage.transform")
    for name in ("load_obj",):
        setattr(sys.modules["pytorch3d.io"], name, None)
    for name in ("chamfer_distance", "mesh_laplacian_smoothing", "mesh_normal_consistency"):
        setattr(sys.modules["pytorch3d.loss"], name, None)
    for name in ("rotate", "resize"):
        setattr(sys.modules["skimage.transform"], name, None)
    # lib.pymaf.utils.imutils pulls in cv2/rembg/human_det at import time; only `uncrop` is
    # referenced on import of mesh_util
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _stub("lib.pymaf"); _stub("lib.pymaf.utils"); _stub("lib.pymaf.utils.imutils", uncrop=None)

    import lib.dataset.mesh_util as mu       # noqa: E402  (reference module)

    mu.SMPLX.__init__ = lambda self: setattr(self, "tedra_dir", "/nonexistent")
    import lib.net                           # noqa: E402

    if "lib.net.NormalNet" in sys.modules:
        sys.modules["lib.net.NormalNet"].VGGLoss = lambda *a, **k: None
    from lib.net import HGPIFuNet            # noqa: E402
    from lib.net.MLP import MLP              # noqa: E402
    from lib.net.geometry import index, orthogonal  # noqa: E402
    from lib.common.train_util import query_func    # noqa: E402
    from lib.common.seg3d_lossless import Seg3dLossless  # noqa: E402

    _loaded = SimpleNamespace(HGPIFuNet=HGPIFuNet, MLP=MLP, index=index, orthogonal=orthogonal,
                              query_func=query_func, Seg3dLossless=Seg3dLossless,
                              cal_sdf_batch=mu.cal_sdf_batch, feat_select=mu.feat_select,
                              barycentric=mu.barycentric_coordinates_of_projection, mesh_util=mu)
    return _loaded


def make_cfg(prior_type: str = "icon", use_filter: bool = True):
    """Attribute-style config carrying the fields the path reads,
    values of configs/icon-filter.yaml / pamir.yaml + lib/common/config.py defaults."""
    net = SimpleNamespace(
        mlp_dim=[256, 512, 256, 128, 1], res_layers=[2, 3, 4], num_stack=2, prior_type=prior_type,
        use_filter=use_filter, in_geo=(("normal_F", 3), ("normal_B", 3)),
        in_nml=(("image", 3), ("T_normal_F", 3), ("T_normal_B", 3)),
        smpl_feats=["sdf", "norm", "vis", "cmap"], gtype="HGPIFuNet", norm_mlp="batch",
        hourglass_dim=6, smpl_dim=7, voxel_dim=7,
        norm="group", hg_down="ave_pool", num_hourglass=2, conv1=[7, 2, 1, 3], conv3x3=[3, 1, 1, 1],
        skip_hourglass=False, use_tanh=False, no_residual=False, hg_depth=2, n_aug=3,
        classifierIMF="MultiSegClassifier", N_freqs=10, geo_w=0.1, norm_w=0.1, dice_w=0.1,
        bce_w=1.0, pifu=False, front_losses=[], back_losses=[], fine_part=[])
    if prior_type != "icon":
        net.in_geo = (("image", 3), ("normal_F", 3), ("normal_B", 3))
    return SimpleNamespace(net=net, root="./data/", overfit=False, sdf_clip=5.0, test_mode=True,
                           num_views=1, batch_size=1, gpus=[0], test_gpus=[0], projection_mode="orthogonal",
                           mcube_res=256, clean_mesh=True)


def build_netG(assets, cfg=None):
    """Instantiate the reference HGPIFuNet on CPU, load the synthetic if_regressor checkpoint
    and bind smpl_feat_dict the way filter() does (lib/net/HGPIFuNet.py:236-245)."""
    import torch

    ref = load()
    cfg = cfg or make_cfg(assets.prior_type)
    torch.manual_seed(1993)
    netG = ref.HGPIFuNet(cfg)
    netG.eval()
    sd = {k: torch.from_numpy(v) for k, v in assets.state_dict.items()}
    missing, unexpected = netG.if_regressor.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in m for m in missing), (missing, unexpected)
    netG.smpl_feat_dict = {
        "smpl_verts": torch.from_numpy(assets.smpl_verts), "smpl_faces": torch.from_numpy(assets.smpl_faces),
        "smpl_vis": torch.from_numpy(assets.smpl_vis), "smpl_cmap": torch.from_numpy(assets.smpl_cmap)}
    return netG, cfg
