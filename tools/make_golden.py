"""Generate tests/golden/*.npz by running the REFERENCE's own Python (imported verbatim from
/root/reference through oracle/ref_loader.py) on the seeded synthetic assets.

Runs only in the build container (the reference tree does not travel to the GPU box); the
fixtures are committed.  The three third-party leaves (kaolin point_to_mesh_distance /
check_sign, pytorch3d vertex normals) are bound to the oracle's CPU restatements - everything
above them is reference code: cal_sdf_batch, barycentrics, clipping incl. the tiled cmap
assignment, index/grid_sample, feat_select, MLP, in_cube mask, query_func, Seg3dLossless.

    python tools/make_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from icon_amd import synth  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_loader.load()
    torch.set_num_threads(os.cpu_count())

    # ---- (a) HGPIFuNet.query through query_func on the body mesh, 4096 stratified points ----
    a = synth.make_assets("body")
    netG, cfg = ref_loader.build_netG(a)
    pts = synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], 4096)
    with torch.no_grad():
        occ = ref.query_func(cfg, netG, [T(a.features)], T(pts)[None])[0, 0].numpy()
        sdf, nrm, cm, vis = ref.cal_sdf_batch(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis),
                                              T(pts)[None])
        img = ref.index(T(a.features), T(pts[:, :2].T.copy())[None])[0].numpy()     # [12, N]
    np.savez_compressed(os.path.join(OUT, "query_body_4096.npz"), points=pts, occ=occ,
                        sdf=sdf[0, :, 0].numpy(), norm=nrm[0].numpy(), cmap=cm[0].numpy(),
                        vis=vis[0, :, 0].numpy().astype(np.float32), img_feat=img)
    print("query_body_4096: occ range", occ.min(), occ.max(), "outliers", int((np.abs(sdf[0, :, 0].numpy()) >= 0.05).sum()))

    # affine calibration + non-identity proj_matrix through query_func
    rng = np.random.RandomState(7)
    A = np.eye(4, dtype=np.float32)
    A[:3, :3] += rng.normal(0, 0.05, (3, 3)).astype(np.float32)
    A[:3, 3] = rng.normal(0, 0.02, 3).astype(np.float32)
    with torch.no_grad():
        occ_p = ref.query_func(cfg, netG, [T(a.features)], T(pts[:1024])[None], proj_matrix=T(A)[None])[0, 0].numpy()
    np.savez_compressed(os.path.join(OUT, "query_body_proj_1024.npz"), points=pts[:1024], proj=A, occ=occ_p)

    # ---- (b) MLP.forward alone -------------------------------------------------------------
    x = rng.normal(0, 1, (13, 777)).astype(np.float32)
    with torch.no_grad():
        y = netG.if_regressor(T(x)[None])[0, 0].numpy()
    np.savez_compressed(os.path.join(OUT, "mlp_777.npz"), x=x, y=y)

    # ---- (c) Seg3dLossless, single-level (dense) and the reference's adaptive schedule -------
    with torch.no_grad():
        for name, resolutions in (("dense17", [17]), ("dense33", [33]), ("adaptive_33_65", [33, 65]),
                                  ("adaptive_17_33_65", [17, 33, 65])):
            eng = ref.Seg3dLossless(query_func=ref.query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                    resolutions=resolutions, align_corners=True, balance_value=0.5, faster=True)
            vol = eng(opt=cfg, netG=netG, features=[T(a.features)], proj_matrix=None)
            vol = vol.numpy().astype(np.float32)
            np.savez_compressed(os.path.join(OUT, f"seg3d_body_{name}.npz"), occ=vol,
                                resolutions=np.array(resolutions))
            print(name, vol.shape, "inside voxels", int((vol > 0.5).sum()))

    # ---- (d) small mesh (fast CPU tests): icosphere, 1500 points ---------------------------------
    b = synth.make_assets("ico")
    netB, cfgB = ref_loader.build_netG(b)
    ptsb = synth.stratified_points(b.smpl_verts[0], b.smpl_faces[0], 1500)
    with torch.no_grad():
        occb = ref.query_func(cfgB, netB, [T(b.features)], T(ptsb)[None])[0, 0].numpy()
        sdfb, nrmb, cmb, visb = ref.cal_sdf_batch(T(b.smpl_verts), T(b.smpl_faces), T(b.smpl_cmap), T(b.smpl_vis),
                                                  T(ptsb)[None])
    np.savez_compressed(os.path.join(OUT, "query_ico_1500.npz"), points=ptsb, occ=occb,
                        sdf=sdfb[0, :, 0].numpy(), norm=nrmb[0].numpy(), cmap=cmb[0].numpy(),
                        vis=visb[0, :, 0].numpy().astype(np.float32))

    # ---- (e) PaMIR / PIFu branches (HGPIFuNet.py:348-357): reference index() + reference MLP ------
    # (HGPIFuNet cannot be constructed with prior_type='pamir' here: it needs voxelize_cuda and SMPL
    #  data files, lib/net/HGPIFuNet.py:107-119; the two-line branch is composed from the
    #  reference's own index / MLP / in_cube expressions.)
    for prior in ("pamir", "pifu"):
        c = synth.make_assets("ico", prior_type="pamir")
        feat = synth.make_feature_planes(6 if prior == "pamir" else 12, 128, synth.SEED)
        sd = synth.make_mlp_state_dict(synth.SEED + (1 if prior == "pamir" else 2), sdf_channel=None)
        mlp = ref.MLP(filter_channels=[13, 512, 256, 128, 1], name="if", res_layers=[2, 3, 4], norm="batch", last_op=None)
        mlp.load_state_dict({k: T(v) for k, v in sd.items()}, strict=False)
        mlp.eval()
        p = rng.uniform(-1.05, 1.05, (2000, 3)).astype(np.float32)
        xyz = T(p.T.copy())[None]
        with torch.no_grad():
            in_cube = ((xyz > -1.0) & (xyz < 1.0)).all(dim=1, keepdim=True).float()
            if prior == "pamir":
                lst = [ref.index(T(feat), xyz[:, :2]), ref.index(T(c.vol_feat), xyz)]
            else:
                lst = [ref.index(T(feat), xyz[:, :2]), xyz[:, 2:3]]
            pred = (in_cube * mlp(torch.cat(lst, 1)))[0, 0].numpy()
        # inputs are re-created by the tests from the same synth seeds (tests/common.py: vol_assets)
        np.savez_compressed(os.path.join(OUT, f"query_{prior}_2000.npz"), points=p, occ=pred)
    print("golden fixtures written to", OUT)
    os.system(f"ls -la {OUT}")


def section_g_display():
    """(g) Seg3dLossless.display (lib/common/seg3d_lossless.py:566-581) of a synthetic 33^3 volume, reference run verbatim"""
    import torch
    from oracle import ref_loader
    ref = ref_loader.load()
    res = 33
    rng = np.random.RandomState(0)
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, res)] * 3, indexing="ij")
    vol = (0.5 + 0.8 * (0.6 - np.sqrt((x * 0.9) ** 2 + (y * 0.6) ** 2 + (z * 1.3) ** 2)) + 0.02 * rng.normal(size=(res, res, res))).astype(np.float32)
    r = ref.Seg3dLossless(query_func=None, b_min=[[-1, 1, -1]], b_max=[[1, -1, 1]], resolutions=[17, res], align_corners=True,
                          balance_value=0.5, device="cpu", visualize=False, debug=False, use_cuda_impl=False, faster=True)
    np.savez_compressed(os.path.join(OUT, "display_33.npz"), vol=vol, image=r.display(torch.from_numpy(vol)))


def section_h_adaptive_257():
    """(h) the schedule apps/ICON.py:62-90 builds for mcube_res=256 - Seg3dLossless, resolutions [33,65,129,257],
    faster=True - run verbatim on the synthetic subject.  The 257^3 volume (68 MB) is stored as subsets: the
    stride-4 sub-lattice, three orthogonal mid planes and 60,000 seeded random voxels; plus the number of
    points the reference queried at every level."""
    ref = ref_loader.load()
    a = synth.make_assets("body")
    netG, cfg = ref_loader.build_netG(a)
    counts = []
    orig = ref.query_func

    def counting_query_func(opt, netG, features, points, proj_matrix=None):
        counts.append(int(points.shape[1]))
        return orig(opt, netG, features, points, proj_matrix)
    res = [33, 65, 129, 257]
    with torch.no_grad():
        eng = ref.Seg3dLossless(query_func=counting_query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                resolutions=res, align_corners=True, balance_value=0.5, faster=True)
        vol = eng(opt=cfg, netG=netG, features=[T(a.features)], proj_matrix=None).numpy().astype(np.float32)
    rng = np.random.RandomState(257)
    idx = rng.randint(0, 257 ** 3, 60000).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "seg3d_body_adaptive_257.npz"), resolutions=np.array(res), queries=np.array(counts),
                        sub4=vol[::4, ::4, ::4], plane_z=vol[128], plane_y=vol[:, 128], plane_x=vol[:, :, 128],
                        idx=idx, samples=vol.reshape(-1)[idx], inside=np.int64((vol > 0.5).sum()))
    print("adaptive 257: queries per level", counts, "inside voxels", int((vol > 0.5).sum()))


def section_k_adaptive_513():
    """(k) the schedule of the SHIPPED default mcube_res=512 (configs/icon-filter.yaml:23; apps/ICON.py:62-72 builds
    [33,65,129,257,513]) - Seg3dLossless, faster=True, run verbatim on the synthetic subject.  The 513^3 volume (540 MB) is
    stored as subsets: the stride-8 sub-lattice, three orthogonal mid planes, 60,000 seeded random voxels and 60,000 seeded
    voxels of the level-set band (0.1 < v < 0.9); plus the number of points the reference queried at every level."""
    ref = ref_loader.load()
    a = synth.make_assets("body")
    netG, cfg = ref_loader.build_netG(a)
    counts = []
    orig = ref.query_func

    def counting_query_func(opt, netG, features, points, proj_matrix=None):
        counts.append(int(points.shape[1]))
        return orig(opt, netG, features, points, proj_matrix)
    res = [33, 65, 129, 257, 513]
    with torch.no_grad():
        eng = ref.Seg3dLossless(query_func=counting_query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                resolutions=res, align_corners=True, balance_value=0.5, faster=True)
        vol = eng(opt=cfg, netG=netG, features=[T(a.features)], proj_matrix=None).numpy().astype(np.float32)
    assert vol.shape == (513, 513, 513)
    rng = np.random.RandomState(513)
    idx = rng.randint(0, 513 ** 3, 60000).astype(np.int64)
    flat = vol.reshape(-1)
    band = np.flatnonzero((flat > 0.1) & (flat < 0.9))
    band_idx = np.sort(rng.choice(band, 60000, replace=False)).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "seg3d_body_adaptive_513.npz"), resolutions=np.array(res), queries=np.array(counts),
                        sub8=vol[::8, ::8, ::8], plane_z=vol[256], plane_y=vol[:, 256], plane_x=vol[:, :, 256],
                        idx=idx, samples=flat[idx], band_idx=band_idx, band_samples=flat[band_idx],
                        inside=np.int64((vol > 0.5).sum()), band=np.int64(band.size),
                        vol_sum=np.float64(flat.astype(np.float64).sum()))
    print("adaptive 513: queries per level", counts, "inside voxels", int((vol > 0.5).sum()), "band", band.size)


def section_i_variants():
    """(i) the configurations outside configs/*.yaml that the reference's classes also build (tests/common.py VARIANTS): smpl_feats
    subsets with and without 'vis', norm_mlp 'weight' / 'group' / 'instance', last_op Sigmoid - the reference's own
    HGPIFuNet.query through query_func with its own MLP class, on 3000 seeded points of the synthetic subject"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import VARIANTS, variant_state_dict
    ref = ref_loader.load()
    a = synth.make_assets("body")
    netG, cfg = ref_loader.build_netG(a)
    pts = synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], 3000, seed=41)
    pts = np.concatenate([pts, np.array([[1.0, 0.2, 0.1], [1.2, 0.0, 0.0]], np.float32)])
    out = {"points": pts}
    saved = (netG.smpl_feats, netG.if_regressor)
    for name, (planes, feats, norm, last_op) in VARIANTS.items():
        c0, sd = variant_state_dict(name, a)
        torch.manual_seed(0)
        mlp = ref.MLP(filter_channels=[c0, 512, 256, 128, 1], name="if", res_layers=[2, 3, 4], norm=norm,
                      last_op=torch.nn.Sigmoid() if last_op == "sigmoid" else None).eval()
        missing, unexpected = mlp.load_state_dict({k: T(v) for k, v in sd.items()}, strict=False)
        assert not unexpected and all("num_batches_tracked" in k for k in missing), (name, missing, unexpected)
        try:
            netG.smpl_feats, netG.if_regressor = feats, mlp
            with torch.no_grad():
                occ = ref.query_func(cfg, netG, [T(a.features[:, :planes])], T(pts)[None])[0, 0].numpy()
        finally:
            netG.smpl_feats, netG.if_regressor = saved
        out["occ_" + name] = occ
        print(f"variant {name}: c0 {c0}, occ range {occ.min():.4f} .. {occ.max():.4f}")
    np.savez_compressed(os.path.join(OUT, "variants.npz"), **out)


def pamir_reference_net(a, ve_gain=1.0):
    """The reference's OWN HGPIFuNet with prior_type='pamir' on CPU: its Voxelization wrapper (lib/net/voxelize.py:64-137) and
    its VolumeEncoder (lib/net/VE.py:114-183) as they are, with what cannot exist here replaced at the LEAF only:
      * read_smpl_constants (asset files, lib/dataset/mesh_util.py:240) -> the synthetic tetrahedralised body's constants;
      * voxelize_cuda.forward_semantic_voxelization (external CUDA wheel, voxelize.py:57) -> the checker's voxeliser
        (oracle/icon_accel.c: orc_semantic_voxelize) behind the wheel's signature - PARITY UNPINNED for this leaf;
      * torch.cuda.FloatTensor / Voxelization.check_input (device assertions, voxelize.py:41-49,191-197) -> CPU.
    Everything between - padding strip, update_param, vertices_to_tetrahedrons, the (b,z,y,x,c) -> (b,c,d,h,w) permute, the
    VolumeEncoder stack, ve(vol, intermediate_output=self.training), index(vol_feat, xyz), the MLP - is reference code."""
    from oracle import oracle as orc
    ref = ref_loader.load()
    vv, tets, code = synth.make_tetra_body(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0])
    faces32 = a.smpl_faces[0].astype(np.int32)
    face_code = (code[faces32[:, 0]] + code[faces32[:, 1]] + code[faces32[:, 2]]) / 3.0
    mod = sys.modules["lib.net.HGPIFuNet"]
    mod.read_smpl_constants = lambda folder: (code, face_code.astype(np.float32), faces32, tets.astype(np.int32))

    def leaf(smpl_vertices, smpl_vertex_code, smpl_tetrahedrons, occ_volume, semantic_volume, weight_sum_volume, sigma):
        # wheel signature (voxelize.py:57-59): surface vertices [B,Vs,3], their codes [B,Vs,3], tetrahedra as POSITIONS [B,T,4,3]
        assert smpl_vertices.shape[0] == 1
        vs = smpl_vertices[0].numpy().astype(np.float32)
        tp = smpl_tetrahedrons[0].numpy().astype(np.float32).reshape(-1, 3)
        allv = np.concatenate([vs, tp], 0)
        tidx = (len(vs) + np.arange(len(tp), dtype=np.int64)).reshape(-1, 4)
        res = semantic_volume.shape[1]
        out = orc.semantic_voxelize(allv, len(vs), smpl_vertex_code[0].numpy().astype(np.float32), tidx, res=res, sigma=float(sigma))
        semantic_volume.copy_(torch.from_numpy(out)[None])
        return occ_volume, semantic_volume, weight_sum_volume
    sys.modules["voxelize_cuda"].forward_semantic_voxelization = leaf
    import lib.net.voxelize as vz
    vz.Voxelization.check_input = lambda self, x: None
    torch.cuda.FloatTensor = lambda *shape: torch.zeros(*shape, dtype=torch.float32)
    cfg = ref_loader.make_cfg("pamir")
    torch.manual_seed(1993)
    netG = ref.HGPIFuNet(cfg)
    netG.eval()
    netG.voxelization.device = torch.device("cpu")
    # the constructor's weights are xavier with gain 0.02 (VE.py:27-52): volume features of 1e-6 would pin nothing - the
    # reference's own initialiser with gain 1, and BatchNorm statistics that are not the identity
    netG.ve.init_weights(init_type="xavier", gain=ve_gain)
    for r in (netG.ve.res0, netG.ve.res1):
        r.init_weights(init_type="xavier", gain=ve_gain)
    g = torch.Generator().manual_seed(7)
    for m in netG.ve.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    return netG, cfg, (vv, tets, code)


def section_j_pamir_real_ve():
    """cfg 4 with the reference's REAL Voxelization wrapper + VolumeEncoder"""
    ref = ref_loader.load()
    a = synth.make_assets("body", prior_type="pamir")
    netG, cfg, (vv, tets, code) = pamir_reference_net(a)
    feat = synth.make_feature_planes(6, 128, synth.SEED)
    sd = synth.make_mlp_state_dict(synth.SEED + 1, sdf_channel=None)
    missing, unexpected = netG.if_regressor.load_state_dict({k: T(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    pad_v, pad_f = 5, 9
    vverts = np.concatenate([vv, np.zeros((pad_v, 3), np.float32)])[None]
    vfaces = np.concatenate([tets, np.zeros((pad_f, 4), np.int64)])[None]
    netG.smpl_feat_dict = {"voxel_verts": T(vverts), "voxel_faces": T(vfaces), "pad_v_num": torch.tensor([pad_v]), "pad_f_num": torch.tensor([pad_f])}
    rng = np.random.RandomState(11)
    pts = rng.uniform(-1.05, 1.05, (4000, 3)).astype(np.float32)
    with torch.no_grad():
        occ = ref.query_func(cfg, netG, [T(feat)], T(pts)[None])[0, 0].numpy()
        vol = netG.voxelization(T(vverts)[:, :-pad_v])                    # what query() fed the encoder
        vol_feat = netG.ve(vol, intermediate_output=False)[-1]
    ve_sd = {"ve." + k: v.numpy() for k, v in netG.ve.state_dict().items() if "num_batches_tracked" not in k}
    np.savez_compressed(os.path.join(OUT, "query_pamir_real_ve.npz"), points=pts, occ=occ, pad_v=pad_v, pad_f=pad_f,
                        vol_feat_sample=vol_feat[0, :, ::4, ::4, ::4].numpy(), vol_sum=np.float64(vol.double().sum().item()),
                        vol_feat_absmax=np.float32(vol_feat.abs().max().item()), **ve_sd)
    print("query_pamir_real_ve: occ range", occ.min(), occ.max(), "vol_feat |max|", float(vol_feat.abs().max()), "std", float(vol_feat.std()),
          "vol occupied", int((vol.abs().sum(1) > 0).sum()))


if __name__ == "__main__":
    only = [s for s in ("--display", "--adaptive257", "--adaptive513", "--variants", "--pamir-real") if s in sys.argv]
    if not only:
        main()
    if not only or "--display" in only:
        section_g_display()
    if not only or "--adaptive257" in only:
        section_h_adaptive_257()
    if not only or "--adaptive513" in only:
        section_k_adaptive_513()
    if not only or "--variants" in only:
        section_i_variants()
    if not only or "--pamir-real" in only:
        section_j_pamir_real_ve()
