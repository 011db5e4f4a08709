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


def test_reference_lossless_schedule_is_dead_code(ref):
    """Seg3dLossless(faster=False) - the constructor default, lib/common/seg3d_lossless.py:48 - runs _forward (:267-478).
    Under this image's PyTorch (and every one since true division of integer tensors, 1.5) its second level indexes with
    float tensors (:296 coords_accum = coords / stride; :343) and raises: the mode cannot have been used with the pinned
    torch either.  AdaptiveReconEngine refuses faster=False with this reason (tests/test_host.py) instead of running
    _forward_faster under its name; the one level that does run (a single resolution) is the dense evaluation."""
    def qf(points, **kw):                                       # batch_eval hands over [1, N, 3] and expects [1, C, N] (:125-144)
        return ((points ** 2).sum(2).sqrt()[:, None] < 0.6).float()
    kw = dict(query_func=qf, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], align_corners=True, balance_value=0.5)
    with pytest.raises(IndexError, match="indices"):
        ref.Seg3dLossless(resolutions=[17, 33], faster=False, **kw)()
    one = ref.Seg3dLossless(resolutions=[17], faster=False, **kw)()
    fast = ref.Seg3dLossless(resolutions=[17], faster=True, **kw)()
    assert torch.equal(one, fast) and one.shape == (17, 17, 17)


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


def test_reference_voxelization_wrapper_around_the_leaf(ref, monkeypatch):
    """lib/net/voxelize.py:64-137 run verbatim on CPU around a stub of the missing voxelize_cuda leaf: pins what
    the reference hands to the leaf (surface vertices = the first len(vertex_code) of voxel_verts, tetrahedra as
    gathered POSITIONS [B,T,4,3], sigma), that the padding is stripped before (HGPIFuNet.py:316-319), and the
    (b,z,y,x,c) -> (b,c,d,h,w) permutation - i.e. that icon_amd.engine.semantic_voxelization(voxel_verts,
    voxel_tets, vertex_code) is fed and laid out like Voxelization.forward.  The leaf's arithmetic stays
    PARITY UNPINNED (restated in oracle/icon_accel.c)."""
    import sys
    vox_mod = __import__("lib.net.voxelize", fromlist=["Voxelization"])
    a = assets("ico")
    vv, tets, code = synth.make_tetra_body(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0])
    res, sigma, seen = 16, 0.05, {}

    def leaf(verts, vcode, tet_pos, occ, sem, wsum, sg):
        seen.update(verts=verts.numpy().copy(), code=vcode.numpy().copy(), tet_pos=tet_pos.numpy().copy(), sigma=sg,
                    shapes=(tuple(occ.shape), tuple(sem.shape), tuple(wsum.shape)), wsum0=float(wsum.flatten()[0]))
        ns = verts.shape[1]
        allv = np.concatenate([verts[0].numpy(), tet_pos[0].numpy().reshape(-1, 3)])
        t = ns + np.arange(tet_pos.shape[1] * 4, dtype=np.int64).reshape(-1, 4)
        out = orc.semantic_voxelize(allv, ns, vcode[0].numpy(), t, res=sem.shape[1], sigma=sg)
        return occ, torch.from_numpy(out)[None], wsum

    monkeypatch.setattr(sys.modules["voxelize_cuda"], "forward_semantic_voxelization", leaf, raising=False)
    monkeypatch.setattr(torch.cuda, "FloatTensor", torch.FloatTensor)
    monkeypatch.setattr(vox_mod.Voxelization, "check_input", lambda self, x: None)         # "supports only cuda tensors"
    faces = a.smpl_faces[0].astype(np.int32)
    vox = vox_mod.Voxelization(code, code[faces].mean(1), faces, tets.astype(np.int32), volume_res=res, sigma=sigma,
                               smooth_kernel_size=7, batch_size=1, device=torch.device("cpu"))
    # the dataset pads voxel_verts / voxel_faces (TestDataset.py:165-170); query() strips it (HGPIFuNet.py:316-319)
    pad_v, pad_f = 5, 3
    voxel_verts = T(np.concatenate([vv, np.zeros((pad_v, 3), np.float32)]))[None]
    voxel_faces = T(np.concatenate([tets, np.zeros((pad_f, 4), np.int64)]).astype(np.int32))[None]
    vs, fs = voxel_verts[:, :-pad_v, :], voxel_faces[:, :-pad_f, :]
    vox.update_param(batch_size=fs.shape[0], smpl_tetra=fs[0].detach().cpu().numpy())
    vol = vox(vs)
    assert vol.shape == (1, 3, res, res, res)
    assert seen["shapes"] == ((1, res, res, res), (1, res, res, res, 3), (1, res, res, res)) and abs(seen["wsum0"] - 1e-3) < 1e-9
    assert seen["sigma"] == sigma
    assert np.array_equal(seen["verts"][0], vv[: len(code)]) and np.array_equal(seen["code"][0], code)
    assert np.array_equal(seen["tet_pos"][0], vv[tets])
    # what our host function computes from (voxel_verts, voxel_tets, vertex_code), checker standing in for the HIP kernel
    mine = orc.semantic_voxelize(vv, len(code), code, tets, res=res, sigma=sigma)
    assert np.array_equal(vol[0].permute(1, 2, 3, 0).numpy(), mine)
    assert mine.any()


@pytest.fixture(scope="module")
def body_net(ref):
    """the reference HGPIFuNet on the synthetic body (372 M parameters: built once per module)"""
    return ref_loader.build_netG(assets("body"))


@pytest.mark.parametrize("feats,planes", [(["sdf", "vis"], 12), (["sdf", "norm", "vis"], 12), (["sdf", "cmap", "vis"], 12),
                                          (["vis", "cmap", "norm", "sdf"], 12),
                                          (["sdf"], 6), (["sdf"], 12), (["sdf", "norm", "cmap"], 6), (["cmap"], 6)])
def test_smpl_feats_subsets_against_the_reference(ref, body_net, feats, planes):
    """cfg.net.smpl_feats (lib/net/HGPIFuNet.py:301-311, :334-346): the reference's own query() with a subset of the SMPL
    features and a regressor of the matching input width vs the oracle's restatement of the same layout
    [img | sdf | cmap? | norm?]; without 'vis' (configs/train/icon-mvp.yaml:40) img is every feature channel, not the selected half"""
    a = assets("body")
    netG, cfg = body_net
    img = planes // 2 if "vis" in feats else planes
    c0 = img + 1 + (3 if "cmap" in feats else 0) + (3 if "norm" in feats else 0)
    sd = synth.make_mlp_state_dict(synth.SEED + 5, dims=(c0, 512, 256, 128, 1)) if (c0, planes, len(feats)) != (13, 12, 4) else a.state_dict
    features = np.ascontiguousarray(a.features[:, :planes])
    saved = (netG.smpl_feats, netG.if_regressor)
    try:
        netG.smpl_feats = feats
        netG.if_regressor = ref.MLP(filter_channels=[c0, 512, 256, 128, 1], name="if", res_layers=[2, 3, 4], norm="batch", last_op=None).eval()
        netG.if_regressor.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        pts = synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], 2500, seed=31)
        with torch.no_grad():
            want = ref.query_func(cfg, netG, [T(features)], T(pts)[None])[0, 0].numpy()
        orc.set_smpl_feats("cmap" in feats, "norm" in feats, "vis" in feats)
        got, X = orc.query_icon(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], features, orc.Mlp(sd), pts,
                                sdf_clip=a.sdf_clip)
        assert X.shape[1] == c0
        assert np.abs(got - want).max() <= 2e-6
    finally:
        orc.set_smpl_feats(True, True, True)
        netG.smpl_feats, netG.if_regressor = saved


def test_weight_norm_regressor_against_the_reference(ref):
    """norm_mlp: 'weight' (lib/net/MLP.py:42-45,64-65): nn.utils.weight_norm on all but the last layer, no norm layers.  The
    reference's own MLP with random g / v vs the oracle on the same state_dict, and the weights the engine hands to the
    library (engine.effective_filters) vs the ones the module applies"""
    from icon_amd.engine import check_regressor, regressor_state_dict, effective_filters
    torch.manual_seed(7)
    dims = [13, 512, 256, 128, 1]
    mlp = ref.MLP(filter_channels=dims, name="if", res_layers=[2, 3, 4], norm="weight", last_op=None).eval()
    with torch.no_grad():
        for l in range(3):
            mlp.filters[l].weight_g.mul_(torch.empty_like(mlp.filters[l].weight_g).uniform_(0.5, 2.0))
        mlp.filters[3].weight.mul_(0.3)
    assert len(mlp.norms) == 0
    check_regressor(mlp)
    sd = regressor_state_dict(mlp)
    assert "filters.0.weight_g" in sd and "filters.0.weight_v" in sd and "filters.3.weight" in sd and "filters.0.weight" not in sd
    x = np.random.RandomState(5).normal(0, 1, (3000, 13)).astype(np.float32)
    with torch.no_grad():
        want = mlp(T(x.T.copy())[None])[0, 0].numpy()
    got = orc.Mlp({k: v.numpy() for k, v in sd.items()}).forward(x)[:, 0]
    assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    eff = effective_filters(sd)
    assert not any(k.endswith("weight_g") or k.endswith("weight_v") for k in eff)
    for l in range(3):
        assert torch.equal(eff[f"filters.{l}.weight"], mlp.filters[l].weight.detach())      # what the forward pre-hook computed
    assert torch.equal(eff["filters.3.weight"], sd["filters.3.weight"])


@pytest.mark.parametrize("kind", ["group", "instance"])
def test_call_normalised_regressors_against_the_reference(ref, body_net, kind):
    """norm_mlp 'group' (lib/common/config.py:80 - the config default) / 'instance' (lib/net/MLP.py:35-41): the reference's own
    query() with such a regressor vs the oracle's whole-call restatement (oracle.CallNormMlp); the statistics are the call's, so
    the same points asked for in two halves give different values - pinned as well"""
    from icon_amd.engine import check_regressor
    from icon_amd.callnorm import spec_of
    a = assets("body")
    netG, cfg = body_net
    torch.manual_seed(3)
    dims = [13, 512, 256, 128, 1]
    mlp = ref.MLP(filter_channels=dims, name="if", res_layers=[2, 3, 4], norm=kind, last_op=None).eval()
    with torch.no_grad():
        for l in range(4):
            mlp.filters[l].weight.copy_(torch.from_numpy(a.state_dict[f"filters.{l}.weight"]))
            mlp.filters[l].bias.copy_(torch.from_numpy(a.state_dict[f"filters.{l}.bias"]))
        if kind == "group":
            for m in mlp.norms:
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.1)
    check_regressor(mlp)
    spec = spec_of(mlp, None, dims[1:-1])
    assert spec.kind == kind and spec.groups == ([32, 32, 32] if kind == "group" else [512, 256, 128]) and spec.eps == [1e-5] * 3
    assert (spec.gamma[0] is None) == (kind == "instance")
    sd = {k: v.numpy() for k, v in mlp.state_dict().items()}
    assert ("norms.0.weight" in sd) == (kind == "group") and "norms.0.running_mean" not in sd
    omlp = orc.CallNormMlp(sd, kind)
    pts = synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], 3000, seed=33)
    saved = netG.if_regressor
    try:
        netG.if_regressor = mlp
        with torch.no_grad():
            want = ref.query_func(cfg, netG, [T(a.features)], T(pts)[None])[0, 0].numpy()
            half = ref.query_func(cfg, netG, [T(a.features)], T(pts[:1500])[None])[0, 0].numpy()
    finally:
        netG.if_regressor = saved
    got, _ = orc.query_icon_callnorm(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], a.features, omlp, pts, sdf_clip=a.sdf_clip)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= 5e-6 * scale
    got_half, _ = orc.query_icon_callnorm(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], a.features, omlp, pts[:1500],
                                          sdf_clip=a.sdf_clip)
    assert np.abs(got_half - half).max() <= 5e-6 * scale
    assert np.abs(half - want[:1500]).max() > 1e-3          # the population of the call matters


@pytest.mark.parametrize("case", ["pifu_size_mlp", "nineteen_inputs", "tanh", "res_layers_1_3", "dict_group_5_layers"])
def test_composed_path_against_the_reference(ref, body_net, case):
    """icon_amd/composed.py - the operator sequence for regressors the fused kernels do not carry - against the reference's own
    query() on the configurations that need it: a PIFu-size 5-layer MLP, 12 planes + every SMPL feature without 'vis' (19
    inputs), last_op Tanh, other res_layers, a GroupNorm state_dict with 5 layers.  The geometry leaf is injected (the checker's
    cal_sdf here, icon_sdf_query on the GPU); everything else is the code that runs in production."""
    from types import SimpleNamespace
    from icon_amd import composed
    a = assets("body")
    netG, cfg = body_net
    torch.manual_seed(5)
    feats, planes, norm, last_op, res_layers = ["sdf", "norm", "vis", "cmap"], 12, "batch", None, [2, 3, 4]
    dims = [13, 512, 256, 128, 1]
    if case == "pifu_size_mlp":
        dims = [13, 1024, 512, 256, 128, 1]
    elif case == "nineteen_inputs":
        feats, dims = ["sdf", "norm", "cmap"], [19, 512, 256, 128, 1]
    elif case == "tanh":
        last_op = torch.nn.Tanh()
    elif case == "res_layers_1_3":
        res_layers = [1, 3]
    elif case == "dict_group_5_layers":
        dims, norm = [13, 64, 64, 32, 32, 1], "group"
    mlp = ref.MLP(filter_channels=dims, name="if", res_layers=res_layers, norm=norm, last_op=last_op).eval()
    with torch.no_grad():
        for m in mlp.norms:
            if hasattr(m, "running_mean"):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            if getattr(m, "weight", None) is not None:
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
    shapes = [(int(f.weight.shape[0]), int(f.weight.shape[1])) for f in mlp.filters]
    n_img = planes // 2 if "vis" in feats else planes
    reason = composed.unsupported_reason(shapes, res_layers, last_op is None, n_img, dims[0])
    assert reason is not None, case
    pts = synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], 2500, seed=51)
    pts = np.concatenate([pts, np.array([[1.0, 0.2, 0.1], [1.2, 0.0, 0.0]], np.float32)])
    saved = (netG.smpl_feats, netG.if_regressor)
    try:
        netG.smpl_feats, netG.if_regressor = feats, mlp
        with torch.no_grad():
            want = ref.query_func(cfg, netG, [T(a.features)], T(pts)[None])[0, 0].numpy()
    finally:
        netG.smpl_feats, netG.if_regressor = saved

    def checker_leaf(p):
        o = orc.cal_sdf(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], p.numpy())
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in o.items() if k in ("sdf", "norm", "cmap", "vis")}
    eng = SimpleNamespace(prior_type="icon", smpl_feats=tuple(feats), sdf_clip=a.sdf_clip, cmap_mode="reference", res_layers=tuple(res_layers),
                          norm_mlp="group" if case == "dict_group_5_layers" else None, last_op=None)
    reg = {k: v for k, v in mlp.state_dict().items()} if case == "dict_group_5_layers" else mlp
    got = composed.query_composed(eng, [T(a.features)], T(pts.T.copy())[None], torch.eye(4)[None], reg, sdf_query=checker_leaf)[0][0, 0].numpy()
    assert np.abs(got - want).max() <= 5e-6 * max(1.0, float(np.abs(want).max())), case
    # what the kernels DO carry is not sent here
    assert composed.unsupported_reason([(512, 13), (256, 512), (128, 269), (1, 141)], [2, 3, 4], True, 6, 13) is None


def test_attach_reads_the_reference_network(ref):
    """IconQueryEngine.attach() on the reference's REAL HGPIFuNet (no device needed up to the first kernel launch):
    every attribute the engine reads exists with the meaning it assumes, the regressor check accepts the shipped
    configuration and refuses the ones the kernels do not evaluate (lib/net/HGPIFuNet.py:48-165,236-245; MLP.py:8-72)"""
    from common import ATTACH_NET_ATTRS, ATTACH_REGRESSOR_ATTRS, ATTACH_SMPL_KEYS
    from icon_amd.engine import IconQueryEngine, IconAmdError, check_regressor, regressor_state_dict
    a = assets("body")
    netG, cfg = ref_loader.build_netG(a)
    for name in ATTACH_NET_ATTRS:
        assert hasattr(netG, name), name
    for name in ATTACH_REGRESSOR_ATTRS:
        assert hasattr(netG.if_regressor, name), name
    assert set(ATTACH_SMPL_KEYS) <= set(netG.smpl_feat_dict)
    original = netG.query
    # the module constant query() reads (HGPIFuNet.py:32,337-342): shipped False; a truthy one is refused
    import sys
    ref_mod = sys.modules[type(netG).__module__]
    assert ref_mod.maskout is False
    ref_mod.maskout = True
    try:
        with pytest.raises(IconAmdError, match="maskout"):
            IconQueryEngine.attach(netG)
    finally:
        ref_mod.maskout = False
    assert netG.query == original
    eng = IconQueryEngine.attach(netG)
    assert eng.netG is netG and netG.icon_amd_engine is eng and netG.query == eng.query and netG.query != original
    assert eng.prior_type == "icon" == netG.prior_type
    assert eng.sdf_clip == pytest.approx(cfg.sdf_clip / 100.0) == pytest.approx(netG.sdf_clip)       # HGPIFuNet.py:70
    assert eng.res_layers == tuple(netG.if_regressor.res_layers) == (2, 3, 4)
    assert set(netG.smpl_feats) == {"sdf", "norm", "vis", "cmap"}
    # the real if_regressor: eval-mode BatchNorm1d, no last_op (cfg.test_mode) -> accepted, state_dict in the packed layout
    check_regressor(netG.if_regressor)
    sd = regressor_state_dict(netG.if_regressor)
    assert [tuple(sd[f"filters.{l}.weight"].shape) for l in range(4)] == [(512, 13, 1), (256, 512, 1), (128, 269, 1), (1, 141, 1)]
    assert all(f"norms.{l}.running_var" in sd for l in range(3)) and not any("num_batches_tracked" in k for k in sd)
    for k, v in a.state_dict.items():
        assert np.array_equal(sd[k].numpy(), v), k
    # what the kernels do not evaluate is refused, by name
    netG.if_regressor.train()
    with pytest.raises(IconAmdError, match="training mode"):
        check_regressor(netG.if_regressor)
    netG.if_regressor.eval()
    # the reference's own MLP class built the way HGPIFuNet.py:128-133 builds it, in the other configurations that exist upstream
    # (whole networks are 372 M parameters each - the regressor is what is checked)
    dims = [13, 512, 256, 128, 1]
    mlp_group = ref.MLP(filter_channels=dims, name="if", res_layers=[2, 3, 4], norm="group", last_op=None).eval()   # lib/common/config.py:80 - the config default
    check_regressor(mlp_group)                                    # evaluated per call since round 3 (icon_amd/callnorm.py)
    with pytest.raises(IconAmdError, match="norm_mlp"):           # ... but a bare state_dict does not say what its norms are
        check_regressor(dict(mlp_group.state_dict()))
    check_regressor(dict(mlp_group.state_dict()), "group")
    # cfg.test_mode False: last_op = nn.Sigmoid() (HGPIFuNet.py:133) is evaluated - the oracle's restatement against the reference's
    # own MLP module with that last_op, on the synthetic checkpoint
    from icon_amd.engine import regressor_last_op
    mlp_sig = ref.MLP(filter_channels=dims, name="if", res_layers=[2, 3, 4], norm="batch", last_op=torch.nn.Sigmoid()).eval()
    mlp_sig.load_state_dict({k: torch.from_numpy(v) for k, v in a.state_dict.items()}, strict=False)
    check_regressor(mlp_sig)
    assert regressor_last_op(mlp_sig) == "sigmoid"
    x = np.random.RandomState(4).normal(0, 1, (3000, 13)).astype(np.float32)
    with torch.no_grad():
        want = mlp_sig(T(x.T.copy())[None])[0, 0].numpy()
    got = orc.Mlp(a.state_dict, last_op="sigmoid").forward(x)[:, 0]
    assert np.abs(got - want).max() <= 2e-6 and want.min() > 0.0 and want.max() < 1.0
    with pytest.raises(IconAmdError, match="last_op"):
        mlp_sig.last_op = torch.nn.Tanh()
        check_regressor(mlp_sig)
    # SMPL features the reference does not have are refused at attach time (HGPIFuNet.py:301-309)
    netG.smpl_feats = ["sdf", "colour"]
    with pytest.raises(IconAmdError, match="smpl_feats"):
        IconQueryEngine.attach(netG)
    netG.smpl_feats = ["sdf", "norm", "vis", "cmap"]
    # and without a device the first launch fails loudly instead of falling back to anything
    if not torch.cuda.is_available():
        with pytest.raises(IconAmdError):
            netG.query(features=[T(a.features)], points=T(np.zeros((1, 3, 4), np.float32)), calibs=torch.eye(4)[None],
                       regressor=netG.if_regressor)


def test_pamir_fixture_is_the_reference_with_its_real_encoder():
    """tests/golden/query_pamir_real_ve.npz is the reference's own HGPIFuNet(prior_type='pamir').query - its Voxelization
    wrapper (lib/net/voxelize.py:64-137) and its VolumeEncoder (lib/net/VE.py:114-183) as they are, the voxelize_cuda wheel
    replaced by the checker's voxeliser at the leaf (tools/make_golden.py: pamir_reference_net).  Re-run it: same answers.
    And the test-local replica of the encoder (common.volume_encoder_replica, what the GPU box runs where the reference
    tree does not exist) loads the reference encoder's state_dict strictly and computes the same volume features."""
    import sys
    from common import ROOT, golden, volume_encoder_replica
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden as mg
    g = golden("query_pamir_real_ve.npz")
    a = synth.make_assets("body", prior_type="pamir")
    netG, cfg, (vv, tets, code) = mg.pamir_reference_net(a)
    for k, v in netG.ve.state_dict().items():
        if "num_batches_tracked" not in k:
            assert np.array_equal(v.numpy(), g["ve." + k]), k
    rep = volume_encoder_replica().eval()
    rep.load_state_dict(netG.ve.state_dict(), strict=True)
    pad_v, pad_f = int(g["pad_v"]), int(g["pad_f"])
    vverts = torch.from_numpy(np.concatenate([vv, np.zeros((pad_v, 3), np.float32)]))[None]
    vfaces = torch.from_numpy(np.concatenate([tets, np.zeros((pad_f, 4), np.int64)]))[None]
    netG.smpl_feat_dict = {"voxel_verts": vverts, "voxel_faces": vfaces, "pad_v_num": torch.tensor([pad_v]), "pad_f_num": torch.tensor([pad_f])}
    sd = synth.make_mlp_state_dict(synth.SEED + 1, sdf_channel=None)
    netG.if_regressor.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    feat = synth.make_feature_planes(6, 128, synth.SEED)
    ref = ref_loader.load()
    with torch.no_grad():
        occ = ref.query_func(cfg, netG, [torch.from_numpy(feat)], torch.from_numpy(g["points"])[None])[0, 0].numpy()
        netG.voxelization.update_param(batch_size=1, smpl_tetra=tets)
        vol = netG.voxelization(vverts[:, :-pad_v])
        f_ref = netG.ve(vol, intermediate_output=False)[-1]
        f_rep = rep(vol, intermediate_output=False)[-1]
    assert np.abs(occ - g["occ"]).max() <= 1e-6
    assert torch.equal(f_ref, f_rep)
    assert np.abs(f_ref[0, :, ::4, ::4, ::4].numpy() - g["vol_feat_sample"]).max() <= 1e-6
