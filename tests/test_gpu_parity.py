"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle and against the
golden fixtures generated from the reference's own Python.

Bars (DESIGN.md "Parity"):
  * integer / boolean outputs - nearest face index, inside flag, visibility flag, the outlier
    sign list - BIT-EXACT against the oracle (same float32 operation sequence on both sides);
  * sdf / norm / cmap / image features: <= 1e-6 absolute (they are bit-exact in practice);
  * occupancy: <= 1e-4 absolute (north-star tolerance), observed ~1e-6: the only difference is
    the summation order of the MLP dot products and the float64 BatchNorm fold.
"""
import numpy as np
import pytest
import torch

from common import assets, golden, oracle_query, orc, rows16, vol_assets
from icon_amd import synth

pytestmark = pytest.mark.gpu

OCC_TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def make_engine(a, **kw):
    from icon_amd.engine import IconQueryEngine
    eng = IconQueryEngine(prior_type="icon", sdf_clip=a.sdf_clip, **kw)
    eng.set_mesh(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in a.state_dict.items()})
    return eng


@pytest.fixture(scope="module")
def body():
    return assets("body")


@pytest.fixture(scope="module")
def eng_body(body):
    return make_engine(body)


# ---------------------------------------------------------------------------------------------
# geometry leaves
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mesh", ["ico", "body"])
def test_vertex_normals_bitexact(mesh):
    from icon_amd.engine import MeshHandle
    a = assets(mesh)
    h = MeshHandle(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    vn = h.vertex_normals().cpu().numpy()
    ref = orc.vertex_normals(a.smpl_verts[0], a.smpl_faces[0])
    assert np.array_equal(vn.view(np.uint32), ref.view(np.uint32))
    st = h.stats()
    assert st["depth"] <= 26 and st["nodes"] >= 1


@pytest.mark.parametrize("mesh,n", [("ico", 5000), ("body", 6000)])
@pytest.mark.parametrize("search", ["bvh", "brute"])
def test_sdf_query_vs_oracle(mesh, n, search):
    from icon_amd.engine import MeshHandle
    a = assets(mesh)
    pts = synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], n)
    h = MeshHandle(T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    g = {k: v.cpu().numpy() for k, v in h.sdf_query(T(pts), search=search).items()}
    o = orc.cal_sdf(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], pts)
    assert np.array_equal(g["face"], o["idx"]), f"{(g['face'] != o['idx']).sum()} nearest-face mismatches"
    assert np.array_equal(g["inside"], o["inside"])
    assert np.array_equal(g["vis"], o["vis"])
    assert np.array_equal(g["sdf"].view(np.uint32), o["sdf"].view(np.uint32)), np.abs(g["sdf"] - o["sdf"]).max()
    assert np.abs(g["norm"] - o["norm"]).max() <= 1e-6
    assert np.abs(g["cmap"] - o["cmap"]).max() <= 1e-6


@pytest.mark.parametrize("n", [1, 63, 4097, 98303, 98304, 200000])
def test_point_search_strategies_agree_with_brute_force(body, n):
    """point mode picks its traversal by batch size (one wavefront per point below 98,304 points,
    Morton-ordered packets above): both must reproduce the brute-force scan exactly, far field and
    points outside the unit cube included"""
    from icon_amd.engine import MeshHandle
    rng = np.random.RandomState(n)
    k = max(n // 2, 1)
    pts = np.concatenate([synth.stratified_points(body.smpl_verts[0], body.smpl_faces[0], k, seed=n),
                          rng.uniform(-1.3, 1.3, (n - k, 3)).astype(np.float32)])[:n]
    rng.shuffle(pts)
    h = MeshHandle(T(body.smpl_verts), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))
    a = h.sdf_query(T(pts), search="bvh")
    b = h.sdf_query(T(pts), search="brute")
    for key in ("face", "inside", "vis", "sdf", "norm", "cmap"):
        assert torch.equal(a[key], b[key]), key


def test_point_search_worst_cases():
    """frontier stress for the one-wavefront-per-point search: the centre of a sphere (every triangle is
    as close as the nearest one, nothing can be pruned), points far outside, points on vertices"""
    from icon_amd.engine import MeshHandle
    ico = assets("ico")
    v = ico.smpl_verts[0]
    c = v.mean(0, keepdims=True).astype(np.float32)
    pts = np.concatenate([c, c + 1e-4, np.array([[5, 5, 5], [-7, 0.1, 0.2], [0, 0, 30]], np.float32), v[:40]]).astype(np.float32)
    h = MeshHandle(T(ico.smpl_verts), T(ico.smpl_faces), T(ico.smpl_cmap), T(ico.smpl_vis))
    a = h.sdf_query(T(pts), search="bvh")
    b = h.sdf_query(T(pts), search="brute")
    for key in ("face", "inside", "vis", "sdf", "norm", "cmap"):
        assert torch.equal(a[key], b[key]), key
    # a finer sphere (20,480 faces, 5,120 leaves all inside the bound of its centre)
    v5, f5 = synth.icosphere(5, radius=0.7, center=(0.0, 0.0, 0.0))
    v5 = v5.astype(np.float32); f5 = f5.astype(np.int64)
    vis5, cm5 = synth.make_vis_cmap(v5, f5)
    h5 = MeshHandle(T(v5[None]), T(f5[None]), T(np.asarray(cm5, np.float32).reshape(1, -1, 3)), T(np.asarray(vis5, np.float32).reshape(1, -1, 1)))
    p5 = np.array([[0, 0, 0], [1e-5, -2e-5, 3e-5], [0.69, 0, 0], [3, 3, 3]], np.float32)
    a, b = h5.sdf_query(T(p5), search="bvh"), h5.sdf_query(T(p5), search="brute")
    for key in ("face", "inside", "sdf"):
        assert torch.equal(a[key], b[key]), key


@pytest.mark.parametrize("n", [70000, 150000])      # below / above the switch to Morton-ordered packets
def test_query_large_unordered_batches(body, n):
    """HGPIFuNet.query on an unordered batch large enough for the packet path: equal to the same points
    queried in two halves (cmap_mode='local' makes points independent) and to the oracle on a sample"""
    eng = make_engine(body, cmap_mode="local")
    rng = np.random.RandomState(n)
    pts = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    cal = torch.eye(4, device=dev())[None]
    full = eng.query([T(body.features)], T(pts.T.copy())[None], cal)[0][0, 0]
    h = n // 2
    a = eng.query([T(body.features)], T(pts[:h].T.copy())[None], cal)[0][0, 0]
    b = eng.query([T(body.features)], T(pts[h:].T.copy())[None], cal)[0][0, 0]
    assert torch.equal(full, torch.cat([a, b]))
    sel = rng.choice(n, 3000, replace=False)
    ref, _ = oracle_query(body, pts[sel], cmap_local=True)
    assert np.abs(full.cpu().numpy()[sel] - ref).max() <= OCC_TOL


def test_sdf_bvh_equals_brute_large(body):
    """50k points incl. a dense far field: BVH pruning never changes the argmin / tie rule"""
    from icon_amd.engine import MeshHandle
    rng = np.random.RandomState(3)
    pts = np.concatenate([synth.stratified_points(body.smpl_verts[0], body.smpl_faces[0], 30000, seed=5),
                          rng.uniform(-1.2, 1.2, (20000, 3)).astype(np.float32)])
    h = MeshHandle(T(body.smpl_verts), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))
    a = h.sdf_query(T(pts), search="bvh")
    b = h.sdf_query(T(pts), search="brute")
    for k in ("face", "inside", "vis", "sdf", "norm", "cmap"):
        assert torch.equal(a[k], b[k]), k


def test_sdf_golden_reference(body):
    """against cal_sdf_batch run verbatim (reference python + oracle leaves)"""
    from icon_amd.engine import MeshHandle
    g = golden("query_body_4096.npz")
    h = MeshHandle(T(body.smpl_verts), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))
    o = {k: v.cpu().numpy() for k, v in h.sdf_query(T(g["points"])).items()}
    assert np.abs(o["sdf"] - g["sdf"]).max() <= 1e-6
    assert np.array_equal(o["vis"], g["vis"])
    # unclamped barycentrics extrapolate in the far field (|norm| up to ~25 here): tolerance relative to the value
    assert (np.abs(o["norm"] - g["norm"]) / np.maximum(1.0, np.abs(g["norm"]))).max() <= 1e-5
    assert (np.abs(o["cmap"] - g["cmap"]) / np.maximum(1.0, np.abs(g["cmap"]))).max() <= 1e-5


# ---------------------------------------------------------------------------------------------
# MLP
# ---------------------------------------------------------------------------------------------
PRECISIONS = ["f32", "f16x3"]
# f32: exact-f32 MFMA chain (differs from the float64 oracle by f32 round-off only);
# f16x3: 22-bit split operands on the f16 matrix cores, f32 accumulate
MLP_TOL = {"f32": 1e-5, "f16x3": 2e-5}
# both carry the FLAT 1e-4 bar on every checkpoint scale below
F32_CLASS = ["f32", "f16x3"]


@pytest.mark.parametrize("precision", F32_CLASS)
@pytest.mark.parametrize("sdf_gain", [0.0, 2.0, 8.0])
@pytest.mark.parametrize("learned_std", [0.05, 0.25, 0.5, 1.0, 2.0])
def test_mlp_checkpoint_scale_sweep(learned_std, sdf_gain, precision):
    """The synthetic checkpoint's last layer attenuates the learned part (learned_std = 0.05); a trained
    regressor produces its whole [0,1] output from it.  Sweep that scale (and the sdf skip gain): every
    f32-class precision must hold the north star's 1e-4 ABSOLUTE tolerance without any scaling of the
    bar (lib/net/MLP.py:49-72 is float32 in the reference)."""
    from icon_amd.engine import MlpHandle
    sd = synth.make_mlp_state_dict(seed=synth.SEED + 3, sdf_gain=sdf_gain, learned_std=learned_std)
    x = synth.representative_rows(65536, 13, seed=int(learned_std * 100) + int(sdf_gain))
    ref = orc.Mlp(sd).forward(x, f64=True)[:, 0]
    y = MlpHandle({k: torch.from_numpy(v) for k, v in sd.items()}).forward(T(rows16(x)), precision=precision).cpu().numpy()
    err = np.abs(y - ref)
    print(f"std {learned_std} gain {sdf_gain} {precision}: max {err.max():.2e} p99.9 {np.quantile(err, 0.999):.2e} |ref|max {np.abs(ref).max():.1f}")
    assert err.max() <= OCC_TOL, err.max()


@pytest.mark.parametrize("precision", F32_CLASS)
def test_mlp_init_net_weights(precision):
    """weights as HGPIFuNet.__init__ leaves them (xavier_normal gain 0.02, default BatchNorm;
    lib/net/net_util.py:73-126): tiny weights exercise the per-layer power-of-two scaling of the split operands"""
    from icon_amd.engine import MlpHandle
    sd = synth.make_mlp_state_dict_init_net()
    x = synth.representative_rows(65536, 13, seed=5)
    ref = orc.Mlp(sd).forward(x, f64=True)[:, 0]
    y = MlpHandle({k: torch.from_numpy(v) for k, v in sd.items()}).forward(T(rows16(x)), precision=precision).cpu().numpy()
    assert np.abs(ref).max() > 1e-3          # not a vacuous comparison (outputs are O(0.01) here)
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max(), np.abs(y - ref).max()   # relative: ~2.5e-6 absolute


@pytest.mark.parametrize("layer", [0, 1])
@pytest.mark.parametrize("g", [1e-3, 1e-4, 1e-5, 1e3])
def test_hidden_activations_below_the_f16_normal_range(g, layer):
    """The split-precision MLP carries the hidden activations as f16 pieces: f16's range on the LOW side too.  A checkpoint
    whose BatchNorm gamma / beta of a hidden layer are g = 1e-3 .. 1e-5 with the next layer's weights multiplied by 1 / g is
    the SAME function (LeakyReLU is positively homogeneous, lib/net/MLP.py:49-72) with hidden activations of 1e-5 - f16
    subnormals, a few significant bits - in front of weights of 1e5.  The packer's per-layer activation scale
    (mlp_f16x3.hip: activation_scale) keeps the pieces in the normal range: 1e-4 absolute against the float64 MLP, as for
    the unscaled checkpoint.  g = 1e3: the same on the high side (activations of 1e3-1e4 stay far from 65504)."""
    from icon_amd.engine import MlpHandle
    sd = synth.make_mlp_state_dict(seed=synth.SEED + 7, sdf_gain=8.0, learned_std=0.5)
    sd = {k: v.copy() for k, v in sd.items()}
    sd[f"norms.{layer}.weight"] = (sd[f"norms.{layer}.weight"] * g).astype(np.float32)
    sd[f"norms.{layer}.bias"] = (sd[f"norms.{layer}.bias"] * g).astype(np.float32)
    w = sd[f"filters.{layer + 1}.weight"].copy()
    n_hidden = 512 if layer == 0 else 256                     # layer 2 also takes the raw 13 inputs behind the 256 activations
    w[:, :n_hidden] = (w[:, :n_hidden] / g).astype(np.float32)
    sd[f"filters.{layer + 1}.weight"] = w
    x = synth.representative_rows(65536, 13, seed=11)
    ref = orc.Mlp(sd).forward(x, f64=True)[:, 0]
    for precision in F32_CLASS:
        y = MlpHandle({k: torch.from_numpy(v) for k, v in sd.items()}).forward(T(rows16(x)), precision=precision).cpu().numpy()
        err = np.abs(y - ref)
        print(f"g {g} layer {layer} {precision}: max {err.max():.2e} |ref|max {np.abs(ref).max():.1f}")
        assert err.max() <= OCC_TOL, (precision, err.max())


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("n", [1, 31, 32, 33, 127, 128, 129, 255, 256, 257, 777, 40000])
def test_mlp_forward_vs_oracle(body, n, precision):
    from icon_amd.engine import MlpHandle
    rng = np.random.RandomState(n)
    x = rng.normal(0, 1, (n, 13)).astype(np.float32)
    rows = rows16(x)
    rows[:, 13:] = np.nan if n == 33 else 7.0      # pad slots and the code word must never reach the GEMM
    mlp = MlpHandle({k: torch.from_numpy(v) for k, v in body.state_dict.items()})
    y = mlp.forward(T(rows), precision=precision).cpu().numpy()
    ref = orc.Mlp(body.state_dict).forward(x, f64=True)[:, 0]
    assert np.isfinite(y).all()
    assert np.abs(y - ref).max() <= MLP_TOL[precision], np.abs(y - ref).max()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_mlp_golden_reference(body, precision):
    from icon_amd.engine import MlpHandle
    g = golden("mlp_777.npz")
    mlp = MlpHandle({k: torch.from_numpy(v) for k, v in body.state_dict.items()})
    y = mlp.forward(T(rows16(g["x"].T.copy())), precision=precision).cpu().numpy()
    assert np.abs(y - g["y"]).max() <= MLP_TOL[precision]


def test_mlp_f16x3_error_statistics(body):
    """how far the split-precision path is from float64, on inputs with a wide dynamic range"""
    from icon_amd.engine import MlpHandle
    rng = np.random.RandomState(5)
    x = (rng.normal(0, 1, (200000, 13)) * np.exp(rng.normal(0, 1.5, (200000, 1)))).astype(np.float32)
    mlp = MlpHandle({k: torch.from_numpy(v) for k, v in body.state_dict.items()})
    ref = orc.Mlp(body.state_dict).forward(x, f64=True)[:, 0]
    e32 = np.abs(mlp.forward(T(rows16(x)), precision="f32").cpu().numpy() - ref)
    e16 = np.abs(mlp.forward(T(rows16(x)), precision="f16x3").cpu().numpy() - ref)
    scale = np.maximum(np.abs(ref), 1.0)
    print(f"max/mean |err|/max(1,|ref|): f32 {np.max(e32 / scale):.2e}/{np.mean(e32 / scale):.2e}  "
          f"f16x3 {np.max(e16 / scale):.2e}/{np.mean(e16 / scale):.2e}")
    assert np.max(e16 / scale) <= 2e-5 and np.mean(e16 / scale) <= 2e-6


@pytest.mark.parametrize("precision", PRECISIONS)
def test_mlp_transpose_detecting(precision):
    """asymmetric one-hot weights: any row/column or k-permutation slip in the packed operands
    shows up as a wrong channel being routed to the output"""
    from icon_amd.engine import MlpHandle
    sd = synth.make_mlp_state_dict(seed=77)
    rng = np.random.RandomState(1)
    for l, (co, ci) in enumerate(synth.mlp_layer_shapes()):
        w = np.zeros((co, ci, 1), np.float32)
        for o in range(co):
            w[o, (o * 7 + 3 * l + 1) % ci, 0] = 1.0 + 0.001 * o
        sd[f"filters.{l}.weight"] = w
        sd[f"filters.{l}.bias"] = (0.01 * rng.normal(size=co)).astype(np.float32)
    x = rng.normal(0, 1, (4096, 13)).astype(np.float32)
    y = MlpHandle({k: torch.from_numpy(v) for k, v in sd.items()}).forward(T(rows16(x)), precision=precision).cpu().numpy()
    ref = orc.Mlp(sd).forward(x, f64=True)[:, 0]
    assert np.abs(y - ref).max() <= MLP_TOL[precision], np.abs(y - ref).max()


# ---------------------------------------------------------------------------------------------
# HGPIFuNet.query
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("cmap_mode", ["reference", "local"])
@pytest.mark.parametrize("n", [1, 3, 257, 5000])
def test_query_points_vs_oracle(body, cmap_mode, n, precision):
    eng = make_engine(body, cmap_mode=cmap_mode, precision=precision)
    pts = synth.stratified_points(body.smpl_verts[0], body.smpl_faces[0], n, seed=n)
    out = eng.query([T(body.features)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])
    assert isinstance(out, list) and out[0].shape == (1, 1, n)
    occ = out[0][0, 0].cpu().numpy()
    ref, _ = oracle_query(body, pts, cmap_local=(cmap_mode == "local"))
    assert np.abs(occ - ref).max() <= OCC_TOL, np.abs(occ - ref).max()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_query_golden_reference(body, precision):
    """the committed output of the reference's own query_func -> HGPIFuNet.query"""
    g = golden("query_body_4096.npz")
    from icon_amd.engine import query_func
    from types import SimpleNamespace
    eng_body = make_engine(body, precision=precision)
    occ = query_func(SimpleNamespace(num_views=1), eng_body, [T(body.features)], T(g["points"])[None])
    assert occ.shape == (1, 1, 4096)
    d = np.abs(occ[0, 0].cpu().numpy() - g["occ"])
    assert d.max() <= OCC_TOL, d.max()


def test_query_golden_projection(eng_body, body):
    g = golden("query_body_proj_1024.npz")
    from icon_amd.engine import query_func
    from types import SimpleNamespace
    # (a) proj_matrix applied by query_func with torch (rocBLAS baddbmm), as the reference does
    occ = query_func(SimpleNamespace(num_views=1), eng_body, [T(body.features)], T(g["points"])[None],
                     proj_matrix=T(g["proj"])[None])[0, 0].cpu().numpy()
    # rocBLAS may round the projected coordinates differently from ATen-CPU; a 1-ulp change of a
    # coordinate can flip one outlier flag and thereby shift the reference's tiled cmap
    # assignment for the rest of the call, so compare in 'local' mode robustly and in
    # 'reference' mode through the in-kernel calibration (b)
    # (b) the calibration folded into the kernel: bit-identical projection arithmetic
    out = eng_body.query([T(body.features)], T(g["points"].T.copy())[None], T(g["proj"])[None])[0][0, 0].cpu().numpy()
    assert np.abs(out - g["occ"]).max() <= OCC_TOL
    frac_bad = (np.abs(occ - g["occ"]) > OCC_TOL).mean()
    assert frac_bad <= 0.05, frac_bad


def test_query_ico_small_mesh():
    a = assets("ico")
    g = golden("query_ico_1500.npz")
    eng = make_engine(a)
    occ = eng.query([T(a.features)], T(g["points"].T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0]
    assert np.abs(occ.cpu().numpy() - g["occ"]).max() <= OCC_TOL


@pytest.mark.parametrize("prior", ["pamir", "pifu"])
def test_query_vol_priors(prior):
    from icon_amd.engine import IconQueryEngine
    g = golden(f"query_{prior}_2000.npz")
    feat, vol, sd = vol_assets(prior)
    eng = IconQueryEngine(prior_type=prior)
    eng.set_regressor({k: torch.from_numpy(v) for k, v in sd.items()})
    if vol is not None:
        eng.set_volume_features(T(vol))
    occ = eng.query([T(feat)], T(g["points"].T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
    assert np.abs(occ - g["occ"]).max() <= OCC_TOL
    ref, _ = orc.query_vol(feat, vol, orc.Mlp(sd), g["points"])
    assert np.abs(occ - ref).max() <= OCC_TOL


def test_regressor_cache_invalidation(body):
    """the engine must use the regressor's CURRENT weights"""
    eng = make_engine(body)
    pts = synth.stratified_points(body.smpl_verts[0], body.smpl_faces[0], 512)
    sd = {k: torch.from_numpy(v.copy()) for k, v in body.state_dict.items()}
    eng.set_regressor(sd)
    args = ([T(body.features)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])
    a = eng.query(*args)[0].clone()
    sd["filters.3.bias"].add_(0.25)          # in-place update bumps the tensor version
    b = eng.query(*args)[0]
    inside = (T(pts).abs() < 1).all(1)
    assert torch.allclose((b - a)[0, 0][inside], torch.full_like(a[0, 0][inside], 0.25), atol=1e-6)


def test_errors(body):
    from icon_amd.engine import IconQueryEngine, IconAmdError
    eng = make_engine(body)
    with pytest.raises(IconAmdError):
        eng.query([T(body.features)], torch.zeros(1, 3, 5), torch.eye(4)[None])      # CPU tensors
    with pytest.raises(IconAmdError):
        eng.query([T(body.features)], torch.zeros(2, 3, 5, device=dev()), torch.eye(4, device=dev())[None])
    with pytest.raises(IconAmdError):
        IconQueryEngine(prior_type="icon", smpl_feats=("sdf", "colour"))   # no such SMPL feature
    e2 = IconQueryEngine()
    with pytest.raises(IconAmdError):
        e2.query([T(body.features)], torch.zeros(1, 3, 5, device=dev()), torch.eye(4, device=dev())[None])


# ---------------------------------------------------------------------------------------------
# dense lattice (reconEngine)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("res", [17, 33])
def test_lattice_vs_golden_seg3d(body, res, precision):
    """== the reference's Seg3dLossless with resolutions=[res], run verbatim"""
    g = golden(f"seg3d_body_dense{res}.npz")
    occ = make_engine(body, precision=precision).eval_slab(T(body.features), res, 0, res).cpu().numpy()
    assert occ.shape == (res, res, res)
    assert np.abs(occ - g["occ"]).max() <= OCC_TOL


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("cmap_mode", ["reference", "local"])
def test_lattice_65_vs_oracle(body, cmap_mode, precision):
    res = 65
    eng = make_engine(body, cmap_mode=cmap_mode, precision=precision)
    occ = eng.eval_slab(T(body.features), res, 0, res).cpu().numpy().ravel()
    ref, _ = oracle_query(body, synth.lattice_points(res), cmap_local=(cmap_mode == "local"))
    assert np.abs(occ - ref).max() <= OCC_TOL
    # same thing through the explicit-point API: the lattice kernel generates identical coordinates
    pts = synth.lattice_points(res)
    q = eng.query([T(body.features)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
    assert np.array_equal(q, occ)


def test_lattice_slab_split_equals_single_call(body):
    """Z-slab sharding: 3 'ranks' on one GPU, sign lists exchanged by hand, == the single call"""
    from icon_amd.recon import slab_bounds
    res, world = 33, 3
    feat = T(body.features)
    full = make_engine(body).eval_slab(feat, res, 0, res)
    engines = [make_engine(body) for _ in range(world)]
    lists, counts = [], []
    for r, e in enumerate(engines):
        z0, z1, _ = slab_bounds(res, world, r)
        s, c = e.slab_features(feat, res, z0, z1)
        counts.append(int(c.item())); lists.append(s[: counts[-1]])
    signs = torch.cat(lists).contiguous()
    parts = []
    for r, e in enumerate(engines):
        z0, z1, _ = slab_bounds(res, world, r)
        parts.append(e.slab_finish(res, z0, z1, signs, sum(counts), sum(counts[:r]), device=dev()))
    assert torch.equal(torch.cat(parts), full)
    # the exchanged list is the oracle's list of outlier signs in lattice order
    o = orc.cal_sdf(body.smpl_verts[0], body.smpl_faces[0], body.smpl_cmap[0], body.smpl_vis[0], synth.lattice_points(res))
    exp = np.sign(o["sdf"][np.abs(o["sdf"]) >= np.float32(body.sdf_clip)]).astype(np.int8)
    assert np.array_equal(signs.cpu().numpy(), exp)


@pytest.mark.parametrize("mesh", ["body", "ico"])
def test_host_built_mesh_gives_the_same_answers(mesh):
    """ICON_AMD_MESH_BUILD=host / icon_debug_set_mesh_build(1): the host builder (the checker of the device build,
    tests/test_gpu_mesh_build.py compares their arrays byte for byte) behind the same handle - same query results"""
    from icon_amd import _lib
    from icon_amd.engine import MeshHandle
    a = assets(mesh)
    args = (T(a.smpl_verts), T(a.smpl_faces), T(a.smpl_cmap), T(a.smpl_vis))
    pts = T(synth.stratified_points(a.smpl_verts[0], a.smpl_faces[0], 3000, seed=3))
    dev_built = MeshHandle(*args)
    _lib.lib().icon_debug_set_mesh_build(1)
    try:
        host_built = MeshHandle(*args)
    finally:
        _lib.lib().icon_debug_set_mesh_build(0)
    assert dev_built.stats() == host_built.stats()
    o1, o2 = dev_built.sdf_query(pts), host_built.sdf_query(pts)
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k


def unpack_signs(msg: np.ndarray):
    """[int64 K][2 bits per sign (sign + 1), four to a byte, low bits first] -> int8 [K]"""
    k = int(msg[:8].view(np.int64)[0])
    b = msg[8:8 + (k + 3) // 4].view(np.uint8)
    s = ((b[:, None] >> (2 * np.arange(4, dtype=np.uint8))) & 3).reshape(-1)[:k].astype(np.int8) - 1
    return s


@pytest.mark.parametrize("pieces", [1, 2, 3])
@pytest.mark.parametrize("world", [2, 3, 5])
def test_lattice_slab_split_gathered_messages(body, world, pieces):
    """the single-collective protocol of the multi-GPU path: every 'rank' writes [int64 count][2-bit signs]
    into a fixed-size message, the concatenation (what all_gather returns) goes to phase 2 as it is; a slab may be
    finished in several pieces (the driver gathers the first half while the second is computed)"""
    from icon_amd.recon import slab_bounds
    res = 33
    feat = T(body.features)
    full = make_engine(body).eval_slab(feat, res, 0, res)
    engines = [make_engine(body) for _ in range(world)]
    per = slab_bounds(res, world, 0)[2]
    stride = 8 + ((per * res * res + 3) // 4 + 7) // 8 * 8
    msgs = []
    for r, e in enumerate(engines):
        z0, z1, _ = slab_bounds(res, world, r)
        msg = torch.full((stride,), 77, dtype=torch.int8, device=dev())      # garbage past the count must not matter
        msg[:8] = 0
        if z1 > z0:
            assert e.slab_features(feat, res, z0, z1, msg=msg) is msg
        msgs.append(msg)
    gathered = torch.cat(msgs).contiguous()
    # the messages decode to the oracle's outlier signs in lattice order
    o = orc.cal_sdf(body.smpl_verts[0], body.smpl_faces[0], body.smpl_cmap[0], body.smpl_vis[0], synth.lattice_points(res))
    exp = np.sign(o["sdf"][np.abs(o["sdf"]) >= np.float32(body.sdf_clip)]).astype(np.int8)
    got = np.concatenate([unpack_signs(m.cpu().numpy()) for m in msgs])
    assert np.array_equal(got, exp)
    parts = []
    for r, e in enumerate(engines):
        z0, z1, _ = slab_bounds(res, world, r)
        if z1 > z0:
            out = torch.full((z1 - z0, res, res), float("nan"), device=dev())
            cuts = np.unique(np.linspace(z0, z1, pieces + 1).round().astype(int))
            for za, zb in zip(cuts[:-1], cuts[1:]):
                e.slab_finish_gathered(res, z0, z1, gathered, stride, world, r, out=out, za=int(za), zb=int(zb))
            parts.append(out)
    assert torch.equal(torch.cat(parts), full)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_lattice_257_properties(body, precision):
    """BASELINE.json's full size (257^3 = 16,974,593 points): size-independent properties"""
    res = 257
    feat = T(body.features)
    e_ref = make_engine(body, cmap_mode="reference", precision=precision)
    e_loc = make_engine(body, cmap_mode="local", precision=precision)
    a = e_ref.eval_slab(feat, res, 0, res)
    b = e_loc.eval_slab(feat, res, 0, res)
    assert a.shape == (res, res, res) and torch.isfinite(a).all()
    # in_cube: the outermost lattice shell is exactly +-1 and therefore exactly zero
    for v in (a, b):
        assert (v[0] == 0).all() and (v[-1] == 0).all() and (v[:, 0] == 0).all() and (v[:, -1] == 0).all()
        assert (v[:, :, 0] == 0).all() and (v[:, :, -1] == 0).all()
    # the two cmap modes differ only on clipped points; a stride-8 sub-lattice is the 33^3 lattice
    sub = b[::8, ::8, ::8].contiguous()
    pts = synth.lattice_points(33)
    q = e_loc.query([feat], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].view(33, 33, 33)
    assert torch.equal(sub, q)
    sdf = e_ref._mesh_handle().sdf_query(T(synth.lattice_points(res, 120, 124)))["sdf"].view(4, res, res)
    near = sdf.abs() < body.sdf_clip
    assert near.any() and torch.equal(a[120:124][near], b[120:124][near])
    # slabs are independent of how the z range is cut (local mode)
    c = e_loc.eval_slab(feat, res, 100, 131)
    assert torch.equal(c, b[100:131])
    # a level set exists and hugs the body: inside fraction ~ body volume / 8
    frac = (a > 0.5).float().mean().item()
    assert 0.005 < frac < 0.02, frac


def test_dense_recon_engine_api(body):
    """reconEngine call contract: kwargs, [D,H,W] layout, None when empty, export_mesh"""
    from icon_amd.recon import DenseReconEngine
    from icon_amd.engine import query_func
    from types import SimpleNamespace
    eng = make_engine(body)
    recon = DenseReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                             resolutions=[33, 65], align_corners=True, balance_value=0.5, faster=True).to(dev())
    sdf = recon(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
    assert sdf.shape == (65, 65, 65) and sdf.is_cuda
    assert any("resolutions" in k for k in recon.state_dict())
    verts, faces = recon.export_mesh(sdf)
    assert verts.shape[1] == 3 and faces.shape[1] == 3 and faces.dtype == torch.int64 and len(faces) > 100
    # generic path (non-default box -> coordinates materialised, one query): same numbers here
    recon2 = DenseReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                              resolutions=[65], align_corners=True).to(dev())
    recon2._lattice_fast_path = lambda p: False
    sdf2 = recon2(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
    assert torch.equal(sdf, sdf2)
    # nothing above 0.5 -> None (seg3d_lossless.py:173-177)
    sd = {k: torch.from_numpy(v.copy()) for k, v in body.state_dict.items()}
    sd["filters.3.bias"] -= 10.0
    eng.set_regressor(sd)
    assert recon(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None) is None


def _shard_worker(rank, world, port, cmap_mode, q):
    import os, sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from types import SimpleNamespace
    from icon_amd.engine import query_func
    from icon_amd.recon import DenseReconEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        body = assets("body")
        eng = make_engine(body, cmap_mode=cmap_mode)
        recon = DenseReconEngine(query_func=query_func, resolutions=[33, 65], align_corners=True).to(dev())
        occ = recon(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
        q.put((rank, occ.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cmap_mode", ["reference", "local"])
def test_zslab_sharding_two_ranks_one_gpu(body, cmap_mode):
    """the real sharded path (HIP kernels + sign-list exchange + slab all_gather), two ranks on this
    one GPU over a gloo group: every rank must end up with the single-process volume, bit for bit"""
    import socket
    import torch.multiprocessing as mp
    single = make_engine(body, cmap_mode=cmap_mode).eval_slab(T(body.features), 65, 0, 65).cpu().numpy()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, cmap_mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, occ in res:
        assert np.array_equal(occ, single), rank


def test_mesh_chamfer_vs_reference(body):
    """BASELINE metric, second half: 'mesh Chamfer vs ref' (definition: lib/dataset/Evaluator.py:200-230,
    restated in tests/common.py; units = [-1,1]-cube units x 100).
      (i)  dense GPU field -> mesh  vs  dense reference field (Seg3dLossless, resolutions=[65]) -> mesh: ~0
      (ii) dense GPU field -> mesh  vs  the reference's ADAPTIVE field (resolutions=[33,65], last level
           interpolated, SURVEY.md finding 1) -> mesh: small but non-zero by construction.
    Tolerances: (i) <= 0.01, (ii) <= 1.6 (half the 65^3 voxel size 2/64*100 = 3.1)."""
    from common import chamfer
    from icon_amd.recon import export_mesh_numpy
    res = 65
    occ = make_engine(body).eval_slab(T(body.features), res, 0, res).cpu().numpy()
    adaptive = golden("seg3d_body_adaptive_33_65.npz")["occ"]
    dense_ref, _ = oracle_query(body, synth.lattice_points(res))
    dense_ref = dense_ref.reshape(res, res, res)

    def mesh(vol):
        v, f = export_mesh_numpy(vol, 0.5)
        v = v.numpy().astype(np.float64)
        v = (v - (res - 1) / 2.0) / ((res - 1) / 2.0)          # apps/ICON.py:758-759
        return v.astype(np.float32), f.numpy()

    vg, fg = mesh(occ)
    vr, fr = mesh(dense_ref)
    va, fa = mesh(adaptive)
    c_dense, _ = chamfer(vg, fg, vr, fr, n=20000)
    c_adapt, p2s = chamfer(vg, fg, va, fa, n=20000)
    print(f"chamfer dense-vs-dense {c_dense:.4f}, dense-vs-adaptive {c_adapt:.4f} (p2s {p2s:.4f}), faces {len(fg)}/{len(fa)}")
    assert len(fg) == len(fr)
    assert c_dense <= 0.01
    assert c_adapt <= 1.6


def _canonical_mesh(v, f):
    """order-independent form: vertices sorted lexicographically, faces re-indexed, rotated to start at
    their smallest vertex (winding preserved) and sorted"""
    # vertices at the SAME position (a lattice value exactly at the level puts the crossings of several edges on the lattice
    # point) are one vertex here: their relative order after the sort would be arbitrary
    uniq, inv = np.unique(v, axis=0, return_inverse=True)
    v, order = uniq, np.arange(len(uniq))
    f = inv.reshape(-1)[f]
    rots = np.stack([f, f[:, [1, 2, 0]], f[:, [2, 0, 1]]], 1)                 # the lexicographically smallest rotation (winding kept)
    key = (rots[..., 0] * (len(v) + 1) + rots[..., 1]) * (len(v) + 1) + rots[..., 2]
    f = rots[np.arange(len(f)), np.argmin(key, axis=1)]
    f = f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))]
    return v[order], f


@pytest.mark.parametrize("res", [33, 129])
def test_gpu_marching_cubes_equals_host(body, res):
    """device marching cubes == host marching cubes as sets (same vertices bit for bit, same triangles)"""
    from icon_amd.recon import export_mesh_device, export_mesh_numpy
    occ = make_engine(body).eval_slab(T(body.features), res, 0, res)
    vd, fd = export_mesh_device(occ, 0.5)
    vh, fh = export_mesh_numpy(occ.cpu().numpy(), 0.5)
    assert vd.shape == vh.shape and fd.shape == fh.shape and len(fh) > 100
    a = _canonical_mesh(vd.cpu().numpy(), fd.cpu().numpy())
    b = _canonical_mesh(vh.numpy(), fh.numpy())
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # noise: every ambiguous configuration, still identical and watertight
    gen = torch.Generator(device=occ.device).manual_seed(1993 + res)
    noisy = occ + (torch.rand(occ.shape, device=occ.device, generator=gen) - 0.5) * 0.8
    noisy.view(-1)[::9973] = 0.5                             # lattice values exactly at the level: coincident crossings
    vd, fd = export_mesh_device(noisy, 0.5)
    vh, fh = export_mesh_numpy(noisy.cpu().numpy(), 0.5)
    a, b = _canonical_mesh(vd.cpu().numpy(), fd.cpu().numpy()), _canonical_mesh(vh.numpy(), fh.numpy())
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # nothing above the level
    v0, f0 = export_mesh_device(torch.zeros_like(occ), 0.5)
    assert len(v0) == 0 and len(f0) == 0


def test_adaptive_recon_matches_reference_volume(body):
    """the reference's own coarse-to-fine reconEngine run verbatim on CPU (golden) vs
    AdaptiveReconEngine + HIP query on the GPU: same batches, same order, same volume"""
    from icon_amd.recon import AdaptiveReconEngine
    from icon_amd.engine import query_func
    from types import SimpleNamespace
    g = golden("seg3d_body_adaptive_33_65.npz")
    eng = make_engine(body)
    recon = AdaptiveReconEngine(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                                resolutions=[33, 65], align_corners=True, balance_value=0.5, faster=True).to(dev())
    vol = recon(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
    assert vol.shape == (65, 65, 65)
    d = np.abs(vol.cpu().numpy() - g["occ"])
    assert d.max() <= OCC_TOL, d.max()
    assert recon.last_stats["queries"] == [33 ** 3]      # two levels: coarse lattice queried, last level interpolated
    # the coarsest level through the lattice kernels (default) or through query_func like the other levels: the same bits
    recon.lattice_level0 = False
    assert torch.equal(vol, recon(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None))
    recon.lattice_level0 = True
    # three levels: the middle one queries only the boundary band - against the reference's volume again
    g3 = golden("seg3d_body_adaptive_17_33_65.npz")
    recon3 = AdaptiveReconEngine(faster=True, query_func=query_func, resolutions=[17, 33, 65], align_corners=True).to(dev())
    vol3 = recon3(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
    q = recon3.last_stats["queries"]
    assert q[0] == 17 ** 3 and 0 < q[1] < 33 ** 3 // 2
    d3 = np.abs(vol3.cpu().numpy() - g3["occ"])
    assert d3.max() <= OCC_TOL, (d3.max(), (d3 > OCC_TOL).sum())
    dense = eng.eval_slab(T(body.features), 65, 0, 65)
    agree = ((vol3 > 0.5) == (dense > 0.5)).float().mean().item()
    assert agree > 0.995


def test_adaptive_recon_257_matches_reference_schedule(body):
    """apps/ICON.py:62-90 at mcube_res=256: Seg3dLossless [33,65,129,257], faster=True, run verbatim on the synthetic
    subject by tools/make_golden.py (h).  Same number of queried points at every level, and the same volume on the
    stored subsets (stride-4 sub-lattice, three mid planes, 60,000 random voxels)."""
    from icon_amd.engine import query_func
    from icon_amd.recon import AdaptiveReconEngine
    from types import SimpleNamespace
    g = golden("seg3d_body_adaptive_257.npz")
    eng = make_engine(body)
    ad = AdaptiveReconEngine(faster=True, query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]],
                             resolutions=[int(r) for r in g["resolutions"]], align_corners=True).to(dev())
    vol = ad(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
    assert vol.shape == (257, 257, 257)
    # the ONE-CALL schedule answered (if native_schedule_reason ever starts refusing, the comparison below would silently be
    # the host-driven form's, and the native kernel would only ever be compared with that form - i.e. with itself)
    assert ad.last_stats.get("native") is True, ad.last_stats
    assert ad.last_stats["queries"] == [int(q) for q in g["queries"]]
    v = vol.cpu().numpy()
    assert np.abs(v[::4, ::4, ::4] - g["sub4"]).max() <= OCC_TOL
    assert np.abs(v[128] - g["plane_z"]).max() <= OCC_TOL and np.abs(v[:, 128] - g["plane_y"]).max() <= OCC_TOL
    assert np.abs(v[:, :, 128] - g["plane_x"]).max() <= OCC_TOL
    assert np.abs(v.reshape(-1)[g["idx"]] - g["samples"]).max() <= OCC_TOL
    assert abs(int((v > 0.5).sum()) - int(g["inside"])) <= 8         # voxels within 1e-4 of the level may flip


@pytest.mark.parametrize("res_list", [[33, 65], [17, 33, 65], [33, 65, 129, 257], [9, 17, 33, 65, 129]])
@pytest.mark.parametrize("cmap_mode", ["reference", "local"])
def test_native_schedule_equals_host_driven_schedule(body, res_list, cmap_mode):
    """icon_adaptive_eval (csrc/adaptive.hip: the whole coarse-to-fine schedule of lib/common/seg3d_lossless.py:152-265 as
    kernels on the stream) against the host-driven schedule (torch bookkeeping around HIP queries, the form pinned against
    the reference's own volumes above): the same points per level, the same volume"""
    from icon_amd.engine import query_func
    from icon_amd.recon import AdaptiveReconEngine
    from types import SimpleNamespace
    eng = make_engine(body, cmap_mode=cmap_mode)
    kw = dict(query_func=query_func, b_min=[[-1.0, 1.0, -1.0]], b_max=[[1.0, -1.0, 1.0]], resolutions=res_list, align_corners=True, faster=True)
    call = dict(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(body.features)], proj_matrix=None)
    nat = AdaptiveReconEngine(**kw).to(dev())
    host = AdaptiveReconEngine(**kw).to(dev())
    host.native = False
    v1 = nat(**call)
    assert nat.last_stats.get("native") is True, nat.last_stats
    v2 = host(**call)
    assert host.last_stats["native"] is False
    assert nat.last_stats["queries"] == host.last_stats["queries"], (nat.last_stats, host.last_stats)
    d = (v1 - v2).abs().max().item()
    print(f"{res_list} {cmap_mode}: queries {nat.last_stats['queries']}, max |native - host-driven| = {d:.3e}")
    assert d <= 1e-6
    v3 = nat(**call)                                   # buffers reused: the same bits again
    assert torch.equal(v1, v3)


def test_fresh_meshes_and_schedules_repeat_bit_for_bit(body):
    """a new device mesh build + the native schedule, 40 times: the build's queues / atomics and the search's shared walks are
    timing dependent in HOW they get there - every volume and every per-level count must equal the first (tools/stress_adaptive.py
    runs 300 per process)"""
    eng = make_engine(body)
    feat = T(body.features)
    v, f, cm, vs = T(body.smpl_verts), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis)
    first = None
    for it in range(40):
        eng.set_mesh(v.clone(), f, cm, vs)
        vol, counts, pos = eng.adaptive_eval(feat, [33, 65, 129, 257])
        if first is None:
            first, c0 = vol.clone(), counts
            assert pos and counts[1] > 0 and counts[2] > 0
        else:
            assert counts == c0 and torch.equal(vol.view(torch.int32), first.view(torch.int32)), f"iteration {it}: {counts} vs {c0}"


def test_native_schedule_returns_none_like_the_reference(body):
    """nothing above 0.5 on the coarsest lattice -> None (seg3d_lossless.py:173-177), native and host-driven alike"""
    import copy
    from icon_amd.engine import query_func
    from icon_amd.recon import AdaptiveReconEngine
    from types import SimpleNamespace
    a = copy.copy(body)
    a.state_dict = {k: v.copy() for k, v in body.state_dict.items()}
    a.state_dict["filters.3.bias"] = a.state_dict["filters.3.bias"] - 100.0
    eng = make_engine(a)
    for native in (True, False):
        r = AdaptiveReconEngine(faster=True, query_func=query_func, resolutions=[17, 33, 65], align_corners=True).to(dev())
        r.native = native
        assert r(opt=SimpleNamespace(num_views=1), netG=eng, features=[T(a.features)], proj_matrix=None) is None


def test_lattice_513_properties(body):
    """cfg 5 size (513^3 = 135,005,697 points, 8.6 GB of MLP input rows): 64-bit indexing, and
    the even sub-lattice IS the 257^3 lattice (identical float coordinates), so in the per-point
    cmap mode the two volumes must agree bit for bit there"""
    feat = T(body.features)
    eng = make_engine(body, cmap_mode="local")
    big = eng.eval_slab(feat, 513, 0, 513)
    assert big.shape == (513, 513, 513) and torch.isfinite(big).all()
    small = eng.eval_slab(feat, 257, 0, 257)
    assert torch.equal(big[::2, ::2, ::2], small)
    assert (big[0] == 0).all() and (big[:, :, -1] == 0).all()
    # reference cmap mode at this size: same numbers off the clipped set, finite everywhere
    ref = make_engine(body, cmap_mode="reference").eval_slab(feat, 513, 250, 262)
    assert torch.isfinite(ref).all()
    sdf = eng._mesh_handle().sdf_query(T(synth.lattice_points(513, 256, 257)))["sdf"].view(513, 513)
    near = sdf.abs() < body.sdf_clip
    assert near.any() and torch.equal(ref[6][near], big[256][near])


# ---------------------------------------------------------------------------------------------
# fused path (features assembled inside the MLP kernel) == materialising path, bit for bit
# ---------------------------------------------------------------------------------------------
def _set_unfused(on):
    import ctypes as C
    from icon_amd import _lib
    _lib.check(_lib.lib().icon_debug_set_unfused(C.c_int(int(on))), "icon_debug_set_unfused")


@pytest.mark.parametrize("cmap_mode", ["reference", "local"])
def test_fused_equals_unfused(body, cmap_mode):
    """precision f16x3 (default) runs k_nearest -> k_sign -> lists -> k_fused_f16x3 with no input rows in HBM;
    forcing the round-1 pipeline (k_features -> X -> patch -> k_mlp_f16x3) must give the same bits: the
    geometry and MLP arithmetic are the same device functions (geom_device.h / mlp_f16x3_device.h)."""
    eng = make_engine(body, cmap_mode=cmap_mode)
    feats = T(body.features)
    eye = torch.eye(4, device=dev())[None]
    cases = [synth.stratified_points(body.smpl_verts[0], body.smpl_faces[0], n, seed=n + 7) for n in (1, 255, 5000, 150000)]
    res = 65
    out = {}
    for mode in (True, False):
        try:
            _set_unfused(mode)
            vol = eng.eval_slab(feats, res, 0, res).clone()
            part = eng.eval_slab(feats, res, 20, 41).clone()
            pts = [eng.query([feats], T(p.T.copy())[None], eye)[0].clone() for p in cases]
        finally:
            _set_unfused(False)
        out[mode] = (vol, part, pts)
    assert torch.equal(out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])
    for a, b in zip(out[True][2], out[False][2]):
        assert torch.equal(a, b)
    ref, _ = oracle_query(body, cases[2], cmap_local=(cmap_mode == "local"))
    assert np.abs(out[False][2][2][0, 0].cpu().numpy() - ref).max() <= OCC_TOL


@pytest.mark.parametrize("prior", ["pamir", "pifu"])
def test_fused_equals_unfused_vol_priors(prior):
    from icon_amd.engine import IconQueryEngine
    feat, vol, sd = vol_assets(prior)
    eng = IconQueryEngine(prior_type=prior)
    eng.set_regressor({k: torch.from_numpy(v) for k, v in sd.items()})
    if vol is not None:
        eng.set_volume_features(T(vol))
    pts = np.random.RandomState(4).uniform(-1.02, 1.02, (3000, 3)).astype(np.float32)
    eye = torch.eye(4, device=dev())[None]
    res = {}
    for mode in (True, False):
        try:
            _set_unfused(mode)
            res[mode] = (eng.query([T(feat)], T(pts.T.copy())[None], eye)[0].clone(), eng.eval_slab(T(feat), 33, 0, 33).clone())
        finally:
            _set_unfused(False)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


# ---------------------------------------------------------------------------------------------
# cfg 4 (pamir.yaml) on the dense lattice
# ---------------------------------------------------------------------------------------------
def test_pamir_lattice_vs_oracle():
    """BASELINE.json configs[3]: the PaMIR prior (image planes [1,6,128,128] + hoisted VolumeEncoder output
    [1,7,32^3], lib/net/HGPIFuNet.py:348-354) over the dense lattice: 65^3 entirely against the checker's
    query_vol, 257^3 on a 60k-point sample of the lattice (points are independent for this prior) plus the
    shell / sub-lattice properties."""
    from icon_amd.engine import IconQueryEngine
    feat, vol, sd = vol_assets("pamir")
    eng = IconQueryEngine(prior_type="pamir")
    eng.set_regressor({k: torch.from_numpy(v) for k, v in sd.items()})
    eng.set_volume_features(T(vol))
    mlp = orc.Mlp(sd)
    occ65 = eng.eval_slab(T(feat), 65, 0, 65).cpu().numpy().ravel()
    ref65, _ = orc.query_vol(feat, vol, mlp, synth.lattice_points(65))
    assert np.abs(occ65 - ref65).max() <= OCC_TOL
    res = 257
    occ = eng.eval_slab(T(feat), res, 0, res)
    assert occ.shape == (res, res, res)
    v = occ.cpu().numpy()
    assert (v[0] == 0).all() and (v[-1] == 0).all() and (v[:, 0] == 0).all() and (v[:, -1] == 0).all() \
        and (v[:, :, 0] == 0).all() and (v[:, :, -1] == 0).all()                       # strict in_cube
    assert np.array_equal(v[::4, ::4, ::4].ravel(), occ65)                             # stride-4 sub-lattice == the 65^3 lattice
    rng = np.random.RandomState(11)
    idx = rng.randint(0, res ** 3, 60000)
    pts = synth.lattice_points(res)[idx]
    ref, _ = orc.query_vol(feat, vol, mlp, pts)
    assert np.abs(v.ravel()[idx] - ref).max() <= OCC_TOL
    split = torch.cat([eng.eval_slab(T(feat), res, 0, 100), eng.eval_slab(T(feat), res, 100, res)])
    assert torch.equal(split, occ)


# ---------------------------------------------------------------------------------------------
# icon-nofilter.yaml: raw normal maps as "features" (use_filter False, lib/net/HGPIFuNet.py:230-233):
# [1,6,H,W] planes -> 3 image channels + 7 SMPL channels = a 10-channel regressor
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", F32_CLASS)
def test_icon_nofilter_layout(body, precision):
    from icon_amd.engine import IconQueryEngine
    dims = [10, 512, 256, 128, 1]
    sd = synth.make_mlp_state_dict(seed=synth.SEED + 9, dims=dims, sdf_channel=3)
    feat = synth.make_feature_planes(6, 96, synth.SEED + 4)           # not 128 x 128: plane size is free
    eng = IconQueryEngine(prior_type="icon", sdf_clip=body.sdf_clip, precision=precision)
    eng.set_mesh(T(body.smpl_verts), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in sd.items()})
    pts = synth.stratified_points(body.smpl_verts[0], body.smpl_faces[0], 3000, seed=77)
    occ = eng.query([T(feat)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
    ref, X = orc.query_icon(body.smpl_verts[0], body.smpl_faces[0], body.smpl_cmap[0], body.smpl_vis[0], feat,
                            orc.Mlp(sd), pts, sdf_clip=body.sdf_clip)
    assert X.shape[1] == 10
    assert np.abs(occ - ref).max() <= OCC_TOL
    res = 33
    vol = eng.eval_slab(T(feat), res, 0, res).cpu().numpy().ravel()
    ref_l, _ = orc.query_icon(body.smpl_verts[0], body.smpl_faces[0], body.smpl_cmap[0], body.smpl_vis[0], feat,
                              orc.Mlp(sd), synth.lattice_points(res), sdf_clip=body.sdf_clip)
    assert np.abs(vol - ref_l).max() <= OCC_TOL


# ---------------------------------------------------------------------------------------------
# edge cases: empty query, degenerate / tiny meshes (the oracle defines the behaviour; both sides agree)
# ---------------------------------------------------------------------------------------------
def test_empty_query_and_tiny_batches(body, eng_body):
    feat = T(body.features)
    eye = torch.eye(4, device=dev())[None]
    out = eng_body.query([feat], torch.zeros((1, 3, 0), device=dev()), eye)
    assert out[0].shape == (1, 1, 0)
    for n in (1, 2, 5):
        pts = synth.stratified_points(body.smpl_verts[0], body.smpl_faces[0], 64, seed=n)[:n]
        occ = eng_body.query([feat], T(pts.T.copy())[None], eye)[0][0, 0].cpu().numpy()
        ref, _ = oracle_query(body, pts)
        assert np.abs(occ - ref).max() <= OCC_TOL


def test_degenerate_and_tiny_meshes_match_the_oracle():
    """zero-area triangles (repeated vertex, collinear corners), a duplicated face, a two-triangle open sheet: nearest
    face / sdf / inside flag are whatever the linear-scan definitions of the checker say - bit for bit"""
    from icon_amd.engine import MeshHandle
    rng = np.random.RandomState(12)
    v = np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0], [0.5, 0.5, 0.1], [0.25, 0.25, 0.3], [0.25, 0.0, 0.0]], np.float32)
    v += rng.normal(0, 1e-3, v.shape).astype(np.float32)
    f = np.array([[0, 1, 2], [1, 3, 2], [0, 1, 1],       # repeated vertex
                  [0, 5, 1],                              # nearly collinear sliver
                  [1, 3, 2],                              # duplicate of face 1: ties -> lowest face index
                  [2, 3, 4], [0, 2, 4]], np.int64)
    cm = rng.rand(len(v), 3).astype(np.float32)
    vis = (rng.rand(len(v)) > 0.5).astype(np.float32)
    pts = rng.uniform(-0.4, 0.9, (4000, 3)).astype(np.float32)
    ref = orc.cal_sdf(v, f, cm, vis, pts)
    mesh = MeshHandle(T(v[None]), T(f[None]), T(cm[None]), T(vis[None, :, None]))
    for search in ("bvh", "brute"):
        got = mesh.sdf_query(T(pts), search=search)
        assert np.array_equal(got["face"].cpu().numpy(), ref["idx"]), search
        assert np.array_equal(got["inside"].cpu().numpy(), ref["inside"]), search
        assert np.array_equal(got["sdf"].cpu().numpy().view(np.int32), ref["sdf"].view(np.int32)), search
        assert np.array_equal(got["vis"].cpu().numpy(), ref["vis"]), search
        ok = np.isfinite(ref["norm"]).all(1)
        assert np.abs(got["norm"].cpu().numpy()[ok] - ref["norm"][ok]).max() <= 1e-5


def test_lattice_rows_with_more_crossings_than_the_row_list_holds(body):
    """12 concentric shells: the lattice rows through the middle are covered by 24 triangles, more than the
    per-row crossing list of the lattice kernels holds (kRowCap = 16, geom_device.h) - those rows take the
    (y,z)-bin scan instead; inside / outside alternates from shell to shell.  Lattice == checker, and the lattice
    kernel == the explicit-point path (which never uses row lists) bit for bit."""
    from types import SimpleNamespace
    v0, f0 = synth.icosphere(1)
    v0 = v0 / np.linalg.norm(v0, axis=1, keepdims=True)
    rng = np.random.RandomState(5)
    vs, fs = [], []
    for k in range(12):
        rot = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        vs.append((v0 @ rot.T) * (0.12 + 0.065 * k) + rng.normal(0, 1e-3, 3))
        fs.append(f0 + k * len(v0))
    verts, faces = np.concatenate(vs).astype(np.float32), np.concatenate(fs).astype(np.int64)
    vis, cmap = synth.make_vis_cmap(verts, faces)
    a = SimpleNamespace(**vars(body))
    a.smpl_verts, a.smpl_faces, a.smpl_vis, a.smpl_cmap = verts[None], faces[None], vis[None], cmap[None]
    res = 33
    pts = synth.lattice_points(res)
    # triangles whose (y,z) projection covers the row y = z = 0 (float64 edge functions)
    t = verts[faces][:, :, 1:].astype(np.float64)
    e = [t[:, i, 0] * t[:, (i + 1) % 3, 1] - t[:, i, 1] * t[:, (i + 1) % 3, 0] for i in range(3)]
    assert int(((np.sign(e[0]) == np.sign(e[1])) & (np.sign(e[1]) == np.sign(e[2]))).sum()) == 24
    eng = make_engine(a)
    occ = eng.eval_slab(T(a.features), res, 0, res).cpu().numpy().ravel()
    ref, _ = oracle_query(a, pts)
    assert np.abs(occ - ref).max() <= OCC_TOL
    q = eng.query([T(a.features)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
    assert np.array_equal(q, occ)


def test_smplx_size_mesh(body):
    """A body with SMPL-X's size (V = 10,242 / F = 20,480 here; SMPL-X: 10,475 / 20,908 - SURVEY.md section 8d): deeper BVH,
    more leaves, larger ray bins.  Nearest face / inside / sdf bit-exact, lattice occupancy <= 1e-4."""
    from types import SimpleNamespace
    from icon_amd.engine import MeshHandle
    v, f = synth.icosphere(5)
    v = v / np.linalg.norm(v, axis=1, keepdims=True)
    rng = np.random.RandomState(31)
    bump = 1.0 + 0.15 * np.sin(5 * v[:, :1]) * np.cos(4 * v[:, 1:2]) + 0.1 * np.sin(7 * v[:, 2:3])
    verts = (v * bump * np.array([0.4, 0.85, 0.3]) + rng.normal(0, 2e-4, v.shape)).astype(np.float32)
    faces = f.astype(np.int64)
    vis, cmap = synth.make_vis_cmap(verts, faces)
    pts = synth.stratified_points(verts, faces, 6000, seed=3)
    ref = orc.cal_sdf(verts, faces, cmap, vis[:, 0], pts)
    mesh = MeshHandle(T(verts[None]), T(faces[None]), T(cmap[None]), T(vis[None]))
    got = mesh.sdf_query(T(pts))
    assert np.array_equal(got["face"].cpu().numpy(), ref["idx"])
    assert np.array_equal(got["inside"].cpu().numpy(), ref["inside"])
    assert np.array_equal(got["sdf"].cpu().numpy().view(np.int32), ref["sdf"].view(np.int32))
    a = SimpleNamespace(**vars(body))
    a.smpl_verts, a.smpl_faces, a.smpl_vis, a.smpl_cmap = verts[None], faces[None], vis[None], cmap[None]
    res = 49
    occ = make_engine(a).eval_slab(T(a.features), res, 0, res).cpu().numpy().ravel()
    ref_occ, _ = oracle_query(a, synth.lattice_points(res))
    assert np.abs(occ - ref_occ).max() <= OCC_TOL


@pytest.mark.parametrize("clip", [0.0, 0.004, 10.0])
def test_unusual_clip_bands(clip):
    """sdf_clip = 0 (every point off the surface is an outlier; a point ON the surface has sign 0), a band thinner than
    the lattice spacing, and a band wider than the cube (no outliers, K = 0): point and lattice mode vs the checker"""
    from types import SimpleNamespace
    a = SimpleNamespace(**vars(assets("ico")))
    a.sdf_clip = clip
    eng = make_engine(a)
    v, f = a.smpl_verts[0], a.smpl_faces[0]
    pts = synth.stratified_points(v, f, 3000, seed=21)
    pts[:len(v)] = v                                              # exactly on the surface: distance 0
    pts[200:400] = v[f[:200]].mean(1).astype(np.float32)          # face centroids (rounded: distance ~1e-8)
    occ = eng.query([T(a.features)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
    ref, _ = oracle_query(a, pts)
    assert np.abs(occ - ref).max() <= OCC_TOL
    res = 33
    vol = eng.eval_slab(T(a.features), res, 0, res).cpu().numpy().ravel()
    ref_l, _ = oracle_query(a, synth.lattice_points(res))
    assert np.abs(vol - ref_l).max() <= OCC_TOL


@pytest.mark.parametrize("n", [1000, 150_000])
def test_non_finite_points_are_far_outside_not_a_fault(body, n):
    """NaN / Inf / 1e30 coordinates (bad caller data; before round 3 a GPU memory fault: the search found no triangle and
    indexed with it): evaluated as far outside the cube - occupancy 0 where the reference returns 0 * NaN - and, in the
    per-point cmap mode, without touching any other point of the call; both search regimes (wave per point / Morton packets).
    Non-finite mesh vertices are refused at mesh creation."""
    from icon_amd.engine import IconQueryEngine, IconAmdError, MeshHandle
    pts = synth.stratified_points(body.smpl_verts[0], body.smpl_faces[0], n, seed=5)
    bad = pts.copy()
    idx = np.arange(0, n, max(n // 50, 1))[:36]
    vals = [np.nan, np.inf, -np.inf, 1e30, -1e30, 3e38]
    for j, i in enumerate(idx):
        bad[i, j % 3] = vals[j % len(vals)]
    eye = torch.eye(4, device=dev())[None]
    for cmap_mode in ("local", "reference"):
        eng = make_engine(body, cmap_mode=cmap_mode)
        clean = eng.query([T(body.features)], T(pts.T.copy())[None], eye)[0][0, 0].cpu().numpy()
        occ = eng.query([T(body.features)], T(bad.T.copy())[None], eye)[0][0, 0].cpu().numpy()
        torch.cuda.synchronize()
        assert (occ[idx] == 0).all() and np.isfinite(occ).all()
        keep = np.ones(n, bool)
        keep[idx] = False
        if cmap_mode == "local":
            assert np.array_equal(occ[keep], clean[keep])
    sdf = eng._mesh_handle().sdf_query(T(bad[idx]))
    torch.cuda.synchronize()
    assert np.isfinite(sdf["sdf"].cpu().numpy()).all() and (sdf["sdf"].cpu().numpy() < 0).all()
    nan_calib = eye.clone()
    nan_calib[0, 0, 0] = float("nan")
    occ = eng.query([T(body.features)], T(pts.T.copy())[None], nan_calib)[0][0, 0].cpu().numpy()
    assert (occ == 0).all()
    v = body.smpl_verts.copy()
    v[0, 17, 1] = np.nan
    with pytest.raises(IconAmdError, match="non-finite"):
        MeshHandle(T(v), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))


def test_sdf_leaf_and_chamfer_in_voxel_units(body):
    """the raw cal_sdf_batch leaf (icon_sdf_query) and metrics.chamfer_p2s take meshes in whatever units the caller has them -
    export_mesh hands out voxel units (0..256), scans come in centimetres: coordinates beyond +-64 must NOT be clamped (round 3
    clamped them in the leaf; only what cannot be searched - NaN / Inf / beyond 1e18 - is changed).  The body scaled by 128
    and shifted by 128: same nearest faces as the unit body, distances x128; Chamfer / P2S of two meshes scale with the units."""
    from icon_amd import metrics
    from icon_amd.engine import MeshHandle
    v1, f = body.smpl_verts[0], body.smpl_faces[0]
    vv = (v1 * 128.0 + 128.0).astype(np.float32)
    pts1 = synth.stratified_points(v1, f, 4000, seed=21)
    pts = (pts1 * 128.0 + 128.0).astype(np.float32)
    h = MeshHandle(T(vv[None]), T(f[None]), T(body.smpl_cmap), T(body.smpl_vis))
    for n in (pts, np.concatenate([pts] * 30)):                    # wave per point, and the Morton-packet path (> 98,304 points)
        g = {k: t.cpu().numpy() for k, t in h.sdf_query(T(n)).items()}
        d2, idx = orc.nearest_brute(vv, f, n[:4000])
        assert np.array_equal(g["face"][:4000], idx)
        want = np.sqrt(d2) / np.sqrt(3.0)
        assert np.abs(np.abs(g["sdf"][:4000]) - want).max() <= 1e-4 * max(1.0, want.max())
        assert np.abs(g["sdf"][:4000]).max() > 64.0 / np.sqrt(3.0) * 0.5      # distances that a +-64 clamp would have cut
    # Chamfer in voxel units == 128 x Chamfer in cube units (same samples: the sampler is scale-free)
    v2 = (v1 * np.float32(1.01)).astype(np.float32)
    c1, p1 = metrics.chamfer_p2s(T(v1), T(f), T(v2), T(f), n=20000)
    c128, p128 = metrics.chamfer_p2s(T(vv), T(f), T((v2 * 128.0 + 128.0).astype(np.float32)), T(f), n=20000)
    assert abs(c128 / c1 - 128.0) <= 0.02 * 128.0 and abs(p128 / p1 - 128.0) <= 0.02 * 128.0, (c1, c128, p1, p128)


@pytest.mark.parametrize("case", ["clip0", "clip_tiny", "clip_huge", "scale3", "scale0.01", "shifted", "outside", "flat", "planes_2x2", "planes_64x96"])
def test_extreme_inputs_vs_oracle(body, case):
    """the corners of the input space: clip bands of 0 / 1e-6 / 100 (every point / no point an outlier), bodies three times
    the cube, a hundredth of it, half outside it, wholly outside it, squashed flat, feature planes of 2x2 and of unequal
    sides; 3^3 .. 33^3 lattices and 1 .. 4097 explicit points against the oracle"""
    from icon_amd.engine import IconQueryEngine
    rs = np.random.RandomState(0)
    clip = {"clip0": 0.0, "clip_tiny": 1e-6, "clip_huge": 100.0}.get(case, body.sdf_clip)
    sc, sh = {"scale3": (3.0, (0, 0, 0)), "scale0.01": (0.01, (0.3, -0.2, 0.1)), "shifted": (1.0, (0.9, 0.0, 0.0)),
              "outside": (1.0, (3.0, 0.0, 0.0)), "flat": ((1.0, 1.0, 1e-4), (0, 0, 0))}.get(case, (1.0, (0, 0, 0)))
    v = (body.smpl_verts * np.asarray(sc, np.float32) + np.asarray(sh, np.float32)).astype(np.float32)
    planes = {"planes_2x2": rs.normal(0, 1, (1, 12, 2, 2)), "planes_64x96": rs.normal(0, 1, (1, 12, 64, 96))}.get(case, body.features)
    planes = np.ascontiguousarray(planes, np.float32)
    omlp = orc.Mlp(body.state_dict)
    for cmap_mode in ("reference", "local"):
        eng = IconQueryEngine(prior_type="icon", sdf_clip=clip, cmap_mode=cmap_mode)
        eng.set_mesh(T(v), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))
        eng.set_regressor({k: torch.from_numpy(x) for k, x in body.state_dict.items()})
        args = (v[0], body.smpl_faces[0], body.smpl_cmap[0], body.smpl_vis[0], planes, omlp)
        for res in (3, 5, 9, 33):
            vol = eng.eval_slab(T(planes), res, 0, res).cpu().numpy().ravel()
            ref, _ = orc.query_icon(*args, synth.lattice_points(res), sdf_clip=clip, cmap_local=(cmap_mode == "local"))
            assert np.abs(vol - ref).max() <= OCC_TOL * max(1.0, np.abs(ref).max()), (case, cmap_mode, res)
        for n in (1, 2, 63, 65, 4097):
            pts = rs.uniform(-1.3, 1.3, (n, 3)).astype(np.float32)
            occ = eng.query([T(planes)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
            ref, _ = orc.query_icon(*args, pts, sdf_clip=clip, cmap_local=(cmap_mode == "local"))
            assert np.abs(occ - ref).max() <= OCC_TOL * max(1.0, np.abs(ref).max()), (case, cmap_mode, n)


@pytest.mark.parametrize("precision", ["f16x3"])
def test_operands_beyond_the_f16_range_are_redone_in_f32(body, precision):
    """the split-precision kernels carry operands as f16 pieces: an input or activation beyond 65504 would be inf and the
    occupancy NaN where the reference's f32 MLP returns a number.  They flag it and the flagged points are recomputed in plain
    f32 (k_rescue_rows / k_rescue_fused): standalone MLP on rows, explicit points and lattices on a body squashed flat (its
    sliver triangles extrapolate |norm| to 1e5), masked points exactly 0"""
    import warnings
    from icon_amd.engine import MlpHandle, IconQueryEngine
    rs = np.random.RandomState(3)
    omlp = orc.Mlp(body.state_dict)
    x = rs.normal(0, 1, (3000, 13)).astype(np.float32)
    big = rs.choice(3000, 40, replace=False)
    x[big, rs.randint(0, 13, 40)] = rs.choice([-1.0, 1.0], 40) * rs.uniform(7e4, 3e6, 40)
    h = MlpHandle({k: torch.from_numpy(v) for k, v in body.state_dict.items()})
    rows = rows16(x)
    rows[:, 13:15] = np.nan                                       # pad slots never reach the arithmetic, rescue included
    got = h.forward(T(rows), precision).cpu().numpy()
    want = omlp.forward(x, f64=True)[:, 0]
    assert np.isfinite(got).all()
    assert (np.abs(got - want) / np.maximum(1.0, np.abs(want))).max() <= OCC_TOL
    assert np.abs(want[big]).max() > 1e3                          # the redone points are the large ones
    v = (body.smpl_verts * np.asarray((1.0, 1.0, 1e-4), np.float32)).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eng = IconQueryEngine(prior_type="icon", sdf_clip=body.sdf_clip, precision=precision)
        eng.set_mesh(T(v), T(body.smpl_faces), T(body.smpl_cmap), T(body.smpl_vis))
        eng.set_regressor({k: torch.from_numpy(w) for k, w in body.state_dict.items()})
        eng._mlp_handle()
        args = (v[0], body.smpl_faces[0], body.smpl_cmap[0], body.smpl_vis[0], body.features, omlp)
        seen_big = 0
        # the f32 arithmetic itself rounds at 2^-24 of the LARGEST operand: the bound carries a term in max |input| (5e-6 of it:
        # just under the range limit the split's low piece no longer reaches 2^-22)
        tol = OCC_TOL
        for res in (5, 9, 17):
            vol = eng.eval_slab(T(body.features), res, 0, res).cpu().numpy().ravel()
            ref, X = orc.query_icon(*args, synth.lattice_points(res), sdf_clip=body.sdf_clip)
            inside = (np.abs(synth.lattice_points(res)) < 1.0).all(1)
            seen_big += int((np.abs(X[inside]).max(1) > 65504).sum())
            assert np.isfinite(vol).all() and (vol[~inside] == 0).all()
            assert (np.abs(vol - ref) <= tol * np.maximum(1.0, np.abs(ref)) + 5e-6 * np.abs(X).max(1)).all(), res
        pts = np.concatenate([synth.lattice_points(9), rs.uniform(-1.2, 1.2, (2000, 3)).astype(np.float32) * np.array([1, 1, 1e-3], np.float32)])
        occ = eng.query([T(body.features)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
        ref, X = orc.query_icon(*args, pts, sdf_clip=body.sdf_clip)
        inside = (np.abs(pts) < 1.0).all(1)
        seen_big += int((np.abs(X[inside]).max(1) > 65504).sum())
        assert np.isfinite(occ).all() and (occ[~inside] == 0).all()
        assert (np.abs(occ - ref) <= tol * np.maximum(1.0, np.abs(ref)) + 5e-6 * np.abs(X).max(1)).all()
    assert seen_big > 0                                           # the case really exercises the range path


@pytest.mark.parametrize("prior", ["pamir", "pifu"])
def test_range_rescue_for_the_volume_priors(prior):
    """feature planes / volume features with a patch of absurd values (3e5): the rows of the points that sample it are beyond
    the f16 range - redone in f32 by k_rescue_fused<pamir|pifu> from the rebuilt rows; explicit points and a lattice"""
    from icon_amd.engine import IconQueryEngine
    rs = np.random.RandomState(4)
    C, Cv = (6, 7) if prior == "pamir" else (12, 1)
    planes = rs.normal(0, 1, (1, C, 128, 128)).astype(np.float32)
    planes[0, 2, 40:60, 40:60] *= 3e5
    vol = rs.normal(0, 1, (1, Cv, 32, 32, 32)).astype(np.float32) if prior == "pamir" else None
    if vol is not None:
        vol[0, 3, 10:14, 10:14, 10:14] *= 2e5
    sd = synth.make_mlp_state_dict(synth.SEED + 9, dims=(13, 512, 256, 128, 1), sdf_channel=None)
    eng = IconQueryEngine(prior_type=prior)
    if prior == "pamir":
        eng.set_volume_features(T(vol))
    eng.set_regressor({k: torch.from_numpy(v) for k, v in sd.items()})
    omlp = orc.Mlp(sd)
    pts = rs.uniform(-1.1, 1.1, (20000, 3)).astype(np.float32)
    occ = eng.query([T(planes)], T(pts.T.copy())[None], torch.eye(4, device=dev())[None])[0][0, 0].cpu().numpy()
    for got, p in ((occ, pts), (eng.eval_slab(T(planes), 33, 0, 33).cpu().numpy().ravel(), synth.lattice_points(33))):
        ref, X = orc.query_vol(planes, vol, omlp, p)
        assert np.isfinite(got).all() and (np.abs(X).max(1) > 65504).sum() > 100
        assert (np.abs(got - ref) <= OCC_TOL * np.maximum(1.0, np.abs(ref)) + 5e-6 * np.abs(X).max(1)).all()
