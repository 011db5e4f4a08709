"""PaMIR semantic voxelisation: the HIP voxeliser against the checker's
restatement, and the hoisted voxelise -> VolumeEncoder -> query() pipeline of the pamir prior.

PARITY UNPINNED for the voxeliser itself: voxelize_cuda (requirements.txt:34) is not under /root/reference and has no
test vectors; the semantics are defined in oracle/icon_accel.c (orc_semantic_voxelize) from the call site
lib/net/voxelize.py:44-59,119-137 and lib/net/HGPIFuNet.py:109-118.  What IS checked: occupancy of the volume ==
inside test of the body mesh (independent code path), HIP == checker (occupancy bit-exact, codes <= 1e-5), value
range and the (z,y,x,c) -> (b,c,d,h,w) layout, and that the engine voxelises + encodes once per image.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from common import assets, orc, vol_assets
from icon_amd import synth


def _tetra():
    a = assets("body")
    return a, synth.make_tetra_body(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0])


def test_oracle_voxelizer_occupancy_is_the_body_interior():
    a, (vv, tets, code) = _tetra()
    res = 32
    out, occ = orc.semantic_voxelize(vv, len(code), code, tets, res=res, sigma=0.05, return_occ=True)
    idx = np.stack(np.meshgrid(*[np.arange(res)] * 3, indexing="ij"), -1).reshape(-1, 3)          # (z, y, x)
    p = ((idx[:, ::-1] + 0.5) / res - 0.5).astype(np.float32)
    inside = orc.Accel(vv[:-1], a.smpl_faces[0]).check_sign(p)                                   # ray parity: orientation-free
    assert (inside == occ.reshape(-1)).mean() > 0.999
    assert occ.any() and (out[~occ] == 0).all()
    assert out[occ].min() >= 0.0 and out[occ].max() <= 1.0                                        # "vol ~ [0,1]" (HGPIFuNet.py:324)
    # Gaussian-weighted average of the surface codes: deep inside it is a blend, at the surface ~ the local code
    near = np.abs(out[occ] - out[occ].mean(0)).max()
    assert near > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("res", [32, 128])
def test_semantic_voxelization_vs_oracle(res):
    from icon_amd.engine import semantic_voxelization
    a, (vv, tets, code) = _tetra()
    dev = torch.device("cuda:0")
    vol = semantic_voxelization(torch.from_numpy(vv)[None].to(dev), torch.from_numpy(tets)[None].to(dev), code, res=res, sigma=0.05)
    assert vol.shape == (1, 3, res, res, res)
    got = vol[0].permute(1, 2, 3, 0).cpu().numpy()                                                # (z, y, x, c)
    ref, occ = orc.semantic_voxelize(vv, len(code), code, tets, res=res, sigma=0.05, return_occ=True)
    assert np.array_equal(np.abs(got).sum(-1) > 0, occ)                                           # same float32 inside test
    assert np.abs(got - ref).max() <= 1e-5
    # padded inputs as the dataset delivers them (TestDataset.py:165-170) give the same volume after stripping
    pad_v = np.concatenate([vv, np.zeros((7, 3), np.float32)])
    vol2 = semantic_voxelization(torch.from_numpy(pad_v)[None].to(dev), torch.from_numpy(tets)[None].to(dev), code, res=res)
    assert torch.equal(vol, vol2)


class _TinyVE(torch.nn.Module):
    """stand-in for lib/net/VE.py:114-183 (3 -> 7 channels, stride-2 3-D convolutions), same call contract"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.net = torch.nn.Sequential(torch.nn.Conv3d(3, 8, 5, 2, 2), torch.nn.ReLU(), torch.nn.Conv3d(8, 7, 3, 2, 1))
        self.calls = 0

    def forward(self, x, intermediate_output=True):
        self.calls += 1
        return [self.net(x)]


@pytest.mark.gpu
def test_pamir_query_with_hoisted_voxelisation_and_volume_encoder():
    """attach() to a network object carrying what HGPIFuNet holds for prior_type='pamir' (voxelization constants, ve,
    smpl_feat_dict with the padded tetra tensors): query() == checker's voxelise -> the same ve on CPU -> query_vol,
    and the voxelise + encode pair runs once per image, not once per query (lib/net/HGPIFuNet.py:314-325 re-runs it)."""
    from icon_amd.engine import IconQueryEngine
    from oracle.query_torch import TorchMLP
    a, (vv, tets, code) = _tetra()
    dev = torch.device("cuda:0")
    feat, _, sd = vol_assets("pamir")
    res_v = 32
    ve = _TinyVE().eval()
    reg = TorchMLP().eval()
    reg.norm, reg.last_op = "batch", None
    reg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    pad_v, pad_f = 5, 9
    vverts = torch.from_numpy(np.concatenate([vv, np.zeros((pad_v, 3), np.float32)]))[None].to(dev)
    vfaces = torch.from_numpy(np.concatenate([tets, np.zeros((pad_f, 4), np.int64)]))[None].to(dev)
    netG = SimpleNamespace(prior_type="pamir", sdf_clip=0.05, smpl_feats=["sdf", "norm", "vis", "cmap"], if_regressor=reg.to(dev),
                           voxelization=SimpleNamespace(smpl_vertex_code=code, volume_res=res_v, sigma=0.05), ve=ve.to(dev),
                           smpl_feat_dict=dict(voxel_verts=vverts, voxel_faces=vfaces,
                                               pad_v_num=torch.tensor([pad_v], device=dev), pad_f_num=torch.tensor([pad_f], device=dev)))
    eng = IconQueryEngine.attach(netG)
    pts = np.random.RandomState(2).uniform(-1.0, 1.0, (4000, 3)).astype(np.float32)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    eye = torch.eye(4, device=dev)[None]
    occ1 = netG.query(features=[T(feat)], points=T(pts.T)[None], calibs=eye, regressor=netG.if_regressor)[0][0, 0].cpu().numpy()
    occ2 = netG.query(features=[T(feat)], points=T(pts.T)[None], calibs=eye, regressor=netG.if_regressor)[0][0, 0].cpu().numpy()
    assert ve.calls == 1 and np.array_equal(occ1, occ2)                         # hoisted: one voxelise + encode per image
    vol_ref = orc.semantic_voxelize(vv, len(code), code, tets, res=res_v, sigma=0.05)             # (z,y,x,c)
    with torch.no_grad():
        vfeat = ve.cpu()(torch.from_numpy(vol_ref).permute(3, 0, 1, 2)[None])[-1].numpy()
    ref, _ = orc.query_vol(feat, vfeat, orc.Mlp(sd), pts)
    assert np.abs(occ1 - ref).max() <= 1e-4
    ve.to(dev)
    netG.smpl_feat_dict["voxel_verts"] = vverts.clone()                         # next image: new tensors -> recomputed
    netG.query(features=[T(feat)], points=T(pts.T)[None], calibs=eye, regressor=netG.if_regressor)
    assert ve.calls == 3        # one CPU call above + the recomputation


@pytest.mark.gpu
def test_pamir_uses_the_reference_voxeliser_when_its_wheel_imports(monkeypatch):
    """voxelizer='auto': with a voxelize_cuda module that has forward_semantic_voxelization the engine calls
    netG.voxelization exactly as HGPIFuNet.query does (update_param + forward on the stripped tensors,
    lib/net/HGPIFuNet.py:316-324) - once per image; voxelizer='reference' without the wheel raises; 'hip' never asks"""
    import sys
    import types
    import warnings
    from icon_amd.engine import IconQueryEngine, IconAmdError, semantic_voxelization
    from oracle.query_torch import TorchMLP
    a, (vv, tets, code) = _tetra()
    dev = torch.device("cuda:0")
    feat, _, sd = vol_assets("pamir")
    res_v = 32
    reg = TorchMLP().eval()
    reg.norm, reg.last_op = "batch", None
    reg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    pad_v, pad_f = 4, 6
    vverts = torch.from_numpy(np.concatenate([vv, np.zeros((pad_v, 3), np.float32)]))[None].to(dev)
    vfaces = torch.from_numpy(np.concatenate([tets, np.zeros((pad_f, 4), np.int64)]))[None].to(dev)

    class RefVox:                                   # stands in for lib.net.voxelize.Voxelization around the wheel
        smpl_vertex_code, volume_res, sigma = code, res_v, 0.05

        def __init__(self):
            self.calls = []

        def update_param(self, batch_size, smpl_tetra):
            self.calls.append(("update_param", batch_size, smpl_tetra.shape))
            self.tets = smpl_tetra

        def __call__(self, verts):
            self.calls.append(("forward", tuple(verts.shape)))
            return semantic_voxelization(verts, torch.from_numpy(self.tets)[None].to(verts.device), self.smpl_vertex_code,
                                         res=self.volume_res, sigma=self.sigma)

    def make_net():
        return SimpleNamespace(prior_type="pamir", sdf_clip=0.05, smpl_feats=["sdf", "norm", "vis", "cmap"], if_regressor=reg.to(dev),
                               voxelization=RefVox(), ve=_TinyVE().eval().to(dev),
                               smpl_feat_dict=dict(voxel_verts=vverts, voxel_faces=vfaces,
                                                   pad_v_num=torch.tensor([pad_v], device=dev), pad_f_num=torch.tensor([pad_f], device=dev)))
    pts = np.random.RandomState(5).uniform(-1.0, 1.0, (1500, 3)).astype(np.float32)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    eye = torch.eye(4, device=dev)[None]

    def run(netG):
        return netG.query(features=[T(feat)], points=T(pts.T)[None], calibs=eye, regressor=netG.if_regressor)[0][0, 0]

    monkeypatch.delitem(sys.modules, "voxelize_cuda", raising=False)
    net_h = make_net(); IconQueryEngine.attach(net_h, voxelizer="hip")
    with warnings.catch_warnings():
        warnings.simplefilter("error")               # 'hip' is silent
        want = run(net_h)
    assert net_h.voxelization.calls == []
    net_r = make_net(); IconQueryEngine.attach(net_r, voxelizer="reference")
    with pytest.raises(IconAmdError, match="voxelize_cuda"):
        run(net_r)
    fake = types.ModuleType("voxelize_cuda"); fake.forward_semantic_voxelization = lambda *a, **k: None
    monkeypatch.setitem(sys.modules, "voxelize_cuda", fake)
    net_a = make_net(); IconQueryEngine.attach(net_a)                                   # voxelizer='auto'
    got = run(net_a); run(net_a)
    assert net_a.voxelization.calls == [("update_param", 1, tuple(tets.shape)), ("forward", (1, len(vv), 3))]   # once per image, stripped tensors
    assert torch.equal(got, want)


@pytest.mark.gpu
def test_pamir_with_the_reference_volume_encoder_stack():
    """cfg 4 end to end with the REAL encoder stack: the fixture is the reference's own HGPIFuNet(prior_type='pamir').query run
    on CPU (tools/make_golden.py section j: its Voxelization wrapper and VolumeEncoder as they are, the voxelize_cuda wheel
    replaced by the checker's voxeliser at the leaf) - here the same voxel tensors, the same encoder weights in a module with
    VE.py's exact layer list (common.volume_encoder_replica; 128^3 -> 32^3 x 7 through MIOpen's k5 / d2 3-D convolutions),
    the HIP voxeliser and the fused query: padding strip, update_param, ve(vol, intermediate_output=False)[-1] as
    lib/net/HGPIFuNet.py:314-325, once per image."""
    from common import golden, volume_encoder_replica
    from icon_amd.engine import IconQueryEngine
    from oracle.query_torch import TorchMLP
    g = golden("query_pamir_real_ve.npz")
    a, (vv, tets, code) = _tetra()
    dev = torch.device("cuda:0")
    feat, _, sd = vol_assets("pamir")
    ve = volume_encoder_replica().eval()
    ve.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ve.")}, strict=False)
    missing = [k for k in ve.state_dict() if "num_batches_tracked" not in k and "ve." + k not in g.files]
    assert not missing, missing
    reg = TorchMLP().eval()
    reg.norm, reg.last_op = "batch", None
    reg.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    pad_v, pad_f = int(g["pad_v"]), int(g["pad_f"])
    vverts = torch.from_numpy(np.concatenate([vv, np.zeros((pad_v, 3), np.float32)]))[None].to(dev)
    vfaces = torch.from_numpy(np.concatenate([tets, np.zeros((pad_f, 4), np.int64)]))[None].to(dev)
    netG = SimpleNamespace(prior_type="pamir", sdf_clip=0.05, smpl_feats=["sdf", "norm", "vis", "cmap"], if_regressor=reg.to(dev),
                           voxelization=SimpleNamespace(smpl_vertex_code=code, volume_res=128, sigma=0.05), ve=ve.to(dev),
                           smpl_feat_dict=dict(voxel_verts=vverts, voxel_faces=vfaces,
                                               pad_v_num=torch.tensor([pad_v], device=dev), pad_f_num=torch.tensor([pad_f], device=dev)))
    eng = IconQueryEngine.attach(netG, voxelizer="hip")
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    eye = torch.eye(4, device=dev)[None]
    pts = g["points"]
    occ = netG.query(features=[T(feat)], points=T(pts.T)[None], calibs=eye, regressor=netG.if_regressor)[0][0, 0].cpu().numpy()
    vf = eng._vol_cached
    assert tuple(vf.shape) == (1, 7, 32, 32, 32)
    dv = np.abs(vf[0, :, ::4, ::4, ::4].cpu().numpy() - g["vol_feat_sample"]).max()
    d = np.abs(occ - g["occ"]).max()
    print(f"real VolumeEncoder stack: max |vol_feat - reference| = {dv:.2e} (|max| {float(g['vol_feat_absmax']):.2f}), max |occ - reference| = {d:.2e}")
    assert dv <= 1e-4 and d <= 1e-4
    netG.query(features=[T(feat)], points=T(pts.T)[None], calibs=eye, regressor=netG.if_regressor)
    assert ve.calls == 1                                                # hoisted: one voxelise + encode per image
