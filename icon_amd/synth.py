"""Seeded synthetic assets for the implicit-surface query path.

The reference ships no weights, no SMPL data and no fixtures (SURVEY.md §0 finding 6), so the
benchmark, the parity tests and smoke() all run on the synthetic inputs specified in
SURVEY.md §8(d):

* a closed, watertight, genus-0 "body" with exactly the SMPL topology size
  (V = 6,890 vertices / F = 13,776 triangles; counts from lib/dataset/TestDataset.py:285),
* ``smpl_vis`` / ``smpl_cmap`` per-vertex attributes (the tensors
  lib/dataset/TestDataset.py:134-148 would produce),
* ``[1, 12, 128, 128]`` hourglass feature planes (lib/net/HGPIFuNet.py:216-228 output shape),
* an ``if_regressor`` checkpoint (state_dict layout of lib/net/MLP.py:26-45) constructed so the
  occupancy field has a 0.5 level set hugging the body and every one of the 13 inputs matters.

Everything is numpy (legacy ``RandomState`` streams are stable across numpy versions) so the
same bytes are produced in the build container and on the GPU box.  Seed 1993 is the
reference's own seed (lib/dataset/TestDataset.py:47).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np

SEED = 1993
SMPL_V, SMPL_F = 6890, 13776
_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


# --------------------------------------------------------------------------------------
# body mesh
# --------------------------------------------------------------------------------------
_CENTER = np.array([0.0, 0.05, 0.0])
_TORSO_AXES = np.array([0.20, 0.38, 0.13])
# limbs: (end point, radius); every capsule starts at _CENTER, so the union is star-shaped
_LIMBS = [
    (np.array([0.00, 0.72, 0.02]), 0.110),    # neck + head
    (np.array([0.40, -0.20, 0.03]), 0.055),   # left arm
    (np.array([-0.40, -0.20, 0.03]), 0.055),  # right arm
    (np.array([0.15, -0.85, 0.00]), 0.075),   # left leg
    (np.array([-0.15, -0.85, 0.00]), 0.075),  # right leg
]


def _fibonacci_sphere(n: int) -> np.ndarray:
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], 1)


def _capsule_exit(dirs: np.ndarray, end: np.ndarray, rad: float) -> np.ndarray:
    """Distance along each unit direction (from the capsule's start point, which is inside it)
    to the capsule boundary; bisection on the convex distance function."""
    seg = end
    seg_len2 = float(seg @ seg)
    lo = np.zeros(len(dirs))
    hi = np.full(len(dirs), np.sqrt(seg_len2) + rad + 1e-3)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        p = dirs * mid[:, None]
        t = np.clip((p @ seg) / seg_len2, 0.0, 1.0)
        d = np.linalg.norm(p - t[:, None] * seg[None, :], axis=1)
        inside = d <= rad
        lo = np.where(inside, mid, lo)
        hi = np.where(inside, hi, mid)
    return 0.5 * (lo + hi)


def _radial(dirs: np.ndarray) -> np.ndarray:
    r = 1.0 / np.sqrt(((dirs / _TORSO_AXES[None, :]) ** 2).sum(1))
    for end, rad in _LIMBS:
        r = np.maximum(r, _capsule_exit(dirs, end - _CENTER, rad))
    return r


def _rotation(rng: np.random.RandomState, max_deg: float) -> np.ndarray:
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(max_deg) * (0.5 + 0.5 * rng.rand())
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def generate_body_mesh(n_verts: int = SMPL_V, seed: int = SEED, n_cand: int = 120_000):
    """Build the star-shaped 'gingerbread' body: union of a torso ellipsoid and five capsules
    radiating from its centre, sampled by farthest-point sampling on the surface and
    triangulated by the convex hull of the sample *directions* (a spherical Delaunay
    triangulation, which the star-shaped radial map carries onto the surface unchanged).
    Returns (verts float32 [V,3], faces int64 [F,3]) with F = 2V-4, outward winding."""
    from scipy.spatial import ConvexHull  # build-time only; the result is committed as .npz

    rng = np.random.RandomState(seed)
    dirs = _fibonacci_sphere(n_cand)
    pts = dirs * _radial(dirs)[:, None]
    # farthest-point sampling on the surface (Euclidean), deterministic start
    chosen = np.empty(n_verts, dtype=np.int64)
    chosen[0] = int(np.argmax(pts[:, 1]))
    dist = np.linalg.norm(pts - pts[chosen[0]], axis=1)
    for k in range(1, n_verts):
        chosen[k] = int(np.argmax(dist))
        dist = np.minimum(dist, np.linalg.norm(pts - pts[chosen[k]], axis=1))
    d = dirs[chosen]
    hull = ConvexHull(d)
    assert len(hull.vertices) == n_verts, "degenerate direction sample"
    faces = hull.simplices.astype(np.int64)
    a, b, c = d[faces[:, 0]], d[faces[:, 1]], d[faces[:, 2]]
    flip = (np.cross(b - a, c - a) * (a + b + c)).sum(1) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]
    # canonical, hull-implementation-independent face order
    faces = np.array([np.roll(f, -int(np.argmin(f))) for f in faces])
    faces = faces[np.lexsort((faces[:, 2], faces[:, 1], faces[:, 0]))]
    verts = pts[chosen] + _CENTER[None, :]
    # move everything off the lattice: a few degrees of rotation and an irrational shift
    R = _rotation(rng, 4.0)
    verts = verts @ R.T + np.array([0.00317, -0.00211, 0.00473])
    assert faces.shape[0] == 2 * n_verts - 4
    return verts.astype(np.float32), faces


def body_mesh_path(n_verts: int = SMPL_V) -> str:
    return os.path.join(_DATA_DIR, f"synth_body_{n_verts}.npz")


def load_body_mesh(n_verts: int = SMPL_V):
    """Load the committed body mesh (generated once by tools/make_synth_body.py)."""
    path = body_mesh_path(n_verts)
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} missing - run `python tools/make_synth_body.py` (needs scipy)")
    z = np.load(path)
    return z["verts"].astype(np.float32), z["faces"].astype(np.int64)


def icosphere(level: int = 2, radius: float = 0.6, center=(0.03, -0.02, 0.01)):
    """Tiny closed mesh for fast unit tests (level 2: 162 verts / 320 faces).  A small fixed
    rotation keeps every vertex off the query lattice."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
         (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
         (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(i, j):
            key = (min(i, j), max(i, j))
            if key not in cache:
                m = v[i] + v[j]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    verts = np.array(v) * radius
    # anisotropic squash so distances are not symmetric, then rotate
    verts = verts * np.array([0.7, 1.2, 0.5])
    R = _rotation(np.random.RandomState(7), 11.0)
    verts = verts @ R.T + np.asarray(center)[None, :]
    return verts.astype(np.float32), np.array(f, dtype=np.int64)


# --------------------------------------------------------------------------------------
# per-vertex attributes
# --------------------------------------------------------------------------------------
def vertex_normals_f64(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    v = verts.astype(np.float64)
    fn = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, faces[:, k], fn)
    return vn / np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-12)


def make_vis_cmap(verts: np.ndarray, faces: np.ndarray):
    """smpl_vis [V,1] in {0,1}: vertex faces the +z camera; smpl_cmap [V,3] in [0,1]:
    bounding-box-normalised position (stand-ins for lib/dataset/TestDataset.py:134-148)."""
    vn = vertex_normals_f64(verts, faces)
    vis = (vn[:, 2:3] > 0.0).astype(np.float32)
    lo, hi = verts.min(0, keepdims=True), verts.max(0, keepdims=True)
    cmap = ((verts - lo) / (hi - lo)).astype(np.float32)
    return vis, cmap


# --------------------------------------------------------------------------------------
# feature planes / volumes
# --------------------------------------------------------------------------------------
def _box_smooth(x: np.ndarray, axes, k: int = 5) -> np.ndarray:
    pad = k // 2
    for ax in axes:
        xp = np.concatenate([np.repeat(np.take(x, [0], ax), pad, ax), x,
                             np.repeat(np.take(x, [-1], ax), pad, ax)], ax)
        cs = np.cumsum(np.concatenate([np.zeros_like(np.take(xp, [0], ax)), xp], ax), ax)
        n = x.shape[ax]
        x = (np.take(cs, np.arange(k, k + n), ax) - np.take(cs, np.arange(0, n), ax)) / k
    return x


def make_feature_planes(channels: int = 12, size: int = 128, seed: int = SEED) -> np.ndarray:
    """[1, C, H, W] float32, N(0,1) smoothed by a 5x5 box then rescaled to unit std per
    channel: smooth but non-constant, so a wrong tap or a swapped half shows up."""
    rng = np.random.RandomState(seed + 11)
    x = rng.normal(size=(channels, size, size))
    x = _box_smooth(x, axes=(1, 2), k=5)
    x = x / x.std(axis=(1, 2), keepdims=True)
    return x[None].astype(np.float32)


def make_feature_volume(channels: int = 7, size: int = 32, seed: int = SEED) -> np.ndarray:
    """[1, C, D, H, W] float32 PaMIR volume-encoder stand-in (lib/net/VE.py:166-183 output)."""
    rng = np.random.RandomState(seed + 23)
    x = rng.normal(size=(channels, size, size, size))
    x = _box_smooth(x, axes=(1, 2, 3), k=3)
    x = x / x.std(axis=(1, 2, 3), keepdims=True)
    return x[None].astype(np.float32)


# --------------------------------------------------------------------------------------
# MLP "checkpoint"
# --------------------------------------------------------------------------------------
MLP_DIMS = [13, 512, 256, 128, 1]   # icon-filter.yaml mlp_dim with element 0 := 13
RES_LAYERS = [2, 3, 4]              # configs/icon-filter.yaml:11
SDF_CHANNEL = 6                     # input order [img0..5, sdf, cmap*3, norm*3]


def mlp_layer_shapes(dims=MLP_DIMS, res_layers=RES_LAYERS):
    """(Cout, Cin) of each Conv1d(k=1), following lib/net/MLP.py:26-33."""
    shapes = []
    for l in range(len(dims) - 1):
        cin = dims[l] + (dims[0] if l in res_layers else 0)
        shapes.append((dims[l + 1], cin))
    return shapes


def make_mlp_state_dict(seed: int = SEED, dims=MLP_DIMS, res_layers=RES_LAYERS,
                        sdf_gain: float = 2.0, learned_std: float = 0.05,
                        sdf_channel: int | None = SDF_CHANNEL) -> dict:
    """state_dict (numpy) with the reference key layout
    ``filters.{l}.weight [Cout,Cin,1]``, ``filters.{l}.bias``,
    ``norms.{l}.{weight,bias,running_mean,running_var}`` (SURVEY.md §5 checkpoint row).

    Hidden layers: U(+-sqrt(6/fan_in)) weights, non-trivial BatchNorm statistics.  The last
    layer is then calibrated on random inputs so that
    ``occ ~= 0.5 + sdf_gain * sdf + learned_std * (unit-variance learned part)``.
    """
    rng = np.random.RandomState(seed + 5)
    shapes = mlp_layer_shapes(dims, res_layers)
    sd = {}
    for l, (co, ci) in enumerate(shapes):
        bound = np.sqrt(6.0 / ci)
        sd[f"filters.{l}.weight"] = rng.uniform(-bound, bound, size=(co, ci, 1)).astype(np.float32)
        sd[f"filters.{l}.bias"] = rng.normal(0, 0.1, size=(co,)).astype(np.float32)
        if l != len(shapes) - 1:
            sd[f"norms.{l}.weight"] = rng.uniform(0.5, 1.5, size=(co,)).astype(np.float32)
            sd[f"norms.{l}.bias"] = rng.normal(0, 0.1, size=(co,)).astype(np.float32)
            sd[f"norms.{l}.running_mean"] = rng.normal(0, 0.1, size=(co,)).astype(np.float32)
            sd[f"norms.{l}.running_var"] = rng.uniform(0.5, 1.5, size=(co,)).astype(np.float32)
    # calibrate the last layer on representative inputs
    n = 4096
    x = rng.normal(0, 1, size=(dims[0], n))
    if sdf_channel is not None:
        x[sdf_channel] = np.where(rng.rand(n) < 0.5, rng.uniform(-0.05, 0.05, n),
                                  np.sign(rng.normal(size=n)))
    last = len(shapes) - 1
    w_last = sd[f"filters.{last}.weight"].astype(np.float64)
    w_last[...] = rng.normal(0, 1, size=w_last.shape)
    sd[f"filters.{last}.weight"] = w_last.astype(np.float32)
    sd[f"filters.{last}.bias"] = np.zeros((dims[-1],), np.float32)
    y = mlp_forward_f64(sd, x, dims, res_layers)
    scale = learned_std / max(float(y.std()), 1e-12)
    w_last = w_last * scale
    if sdf_channel is not None:
        skip0 = shapes[last][1] - dims[0]   # raw input is concatenated last (MLP.py:62)
        w_last[:, skip0 + sdf_channel, 0] = sdf_gain
    sd[f"filters.{last}.weight"] = w_last.astype(np.float32)
    sd[f"filters.{last}.bias"] = np.full((dims[-1],), 0.5 - scale * float(y.mean()), np.float32)
    return sd


def make_mlp_state_dict_init_net(seed: int = SEED, dims=MLP_DIMS, res_layers=RES_LAYERS, init_gain: float = 0.02) -> dict:
    """state_dict as the reference's constructor leaves it: ``init_net(self)`` (lib/net/HGPIFuNet.py:165)
    -> ``init_weights(net, 'xavier', 0.02)`` (lib/net/net_util.py:73-126): Conv1d weights
    xavier_normal with gain 0.02, zero biases; BatchNorm1d keeps torch's defaults (weight 1, bias 0,
    running_mean 0, running_var 1 - only 'BatchNorm2d' is matched by the initialiser)."""
    rng = np.random.RandomState(seed + 7)
    sd = {}
    for l, (co, ci) in enumerate(mlp_layer_shapes(dims, res_layers)):
        std = init_gain * np.sqrt(2.0 / (ci + co))
        sd[f"filters.{l}.weight"] = rng.normal(0, std, size=(co, ci, 1)).astype(np.float32)
        sd[f"filters.{l}.bias"] = np.zeros((co,), np.float32)
        if l != len(dims) - 2:
            sd[f"norms.{l}.weight"] = np.ones((co,), np.float32)
            sd[f"norms.{l}.bias"] = np.zeros((co,), np.float32)
            sd[f"norms.{l}.running_mean"] = np.zeros((co,), np.float32)
            sd[f"norms.{l}.running_var"] = np.ones((co,), np.float32)
    return sd


def representative_rows(n: int, c0: int = 13, seed: int = SEED) -> np.ndarray:
    """[n, c0] float32 MLP inputs with the statistics of the query path (lib/net/HGPIFuNet.py:298-311):
    image features ~ N(0,1); sdf +-1 for ~90 % (clipped outliers) else inside the clip band; cmap the
    outlier signs or [0,1]; norm a unit vector."""
    rng = np.random.RandomState(seed + 31)
    x = rng.normal(0, 1, (n, c0)).astype(np.float32)
    if c0 == 13:
        out = rng.rand(n) < 0.9
        x[:, 6] = np.where(out, np.sign(rng.normal(size=n)), rng.uniform(-0.05, 0.05, n))
        x[:, 7:10] = np.where(out[:, None], np.sign(rng.normal(size=(n, 3))), rng.rand(n, 3))
        v = rng.normal(size=(n, 3))
        x[:, 10:13] = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-6)
    return x.astype(np.float32)


def mlp_forward_f64(sd: dict, x: np.ndarray, dims=MLP_DIMS, res_layers=RES_LAYERS) -> np.ndarray:
    """float64 evaluation of lib/net/MLP.py:49-72 (Conv1d k=1, BatchNorm1d eval with eps 1e-5,
    LeakyReLU 0.01, input re-concatenated before res layers, no last_op in test mode).
    x: [C0, N] -> [C_last, N].  Used for calibration and as the high-precision reference."""
    x = x.astype(np.float64)
    y = x
    n_layers = len(dims) - 1
    for l in range(n_layers):
        inp = np.concatenate([y, x], 0) if l in res_layers else y
        W = sd[f"filters.{l}.weight"][:, :, 0].astype(np.float64)
        y = W @ inp + sd[f"filters.{l}.bias"].astype(np.float64)[:, None]
        if l != n_layers - 1:
            g = sd[f"norms.{l}.weight"].astype(np.float64)[:, None]
            b = sd[f"norms.{l}.bias"].astype(np.float64)[:, None]
            m = sd[f"norms.{l}.running_mean"].astype(np.float64)[:, None]
            v = sd[f"norms.{l}.running_var"].astype(np.float64)[:, None]
            y = (y - m) / np.sqrt(v + 1e-5) * g + b
            y = np.where(y > 0, y, 0.01 * y)
    return y


# --------------------------------------------------------------------------------------
# bundles
# --------------------------------------------------------------------------------------
def make_assets(mesh: str = "body", seed: int = SEED, prior_type: str = "icon") -> SimpleNamespace:
    """All per-image constants of one synthetic subject, numpy, reference tensor shapes."""
    if mesh == "body":
        verts, faces = load_body_mesh()
    elif mesh == "ico":
        verts, faces = icosphere(2)
    elif mesh == "ico3":
        verts, faces = icosphere(3)
    else:
        raise ValueError(mesh)
    vis, cmap = make_vis_cmap(verts, faces)
    a = SimpleNamespace(prior_type=prior_type, seed=seed)
    a.smpl_verts = verts[None]                       # [1,V,3] f32
    a.smpl_faces = faces[None]                       # [1,F,3] i64
    a.smpl_vis = vis[None]                           # [1,V,1] f32
    a.smpl_cmap = cmap[None]                         # [1,V,3] f32
    a.sdf_clip = 5.0 / 100.0                         # lib/common/config.py:36, HGPIFuNet.py:70
    if prior_type == "icon":
        a.features = make_feature_planes(12, 128, seed)
        a.state_dict = make_mlp_state_dict(seed)
    elif prior_type == "pamir":
        a.features = make_feature_planes(6, 128, seed)
        a.vol_feat = make_feature_volume(7, 32, seed)
        a.state_dict = make_mlp_state_dict(seed, sdf_channel=None)
    else:
        raise ValueError(prior_type)
    return a


def make_tetra_body(verts: np.ndarray, faces: np.ndarray, cmap: np.ndarray):
    """Stand-in for the tetrahedralised SMPL of the PaMIR prior (lib/dataset/TestDataset.py:150-192,
    lib/dataset/body_model.py:233-395 - the real tetrahedra need credential-gated data): the surface vertices
    followed by ONE interior vertex, every surface triangle joined to it (the synthetic body is star-shaped
    around its centre), scaled by 0.5 with z flipped exactly as compute_voxel_verts does (:172-179).
    -> voxel_verts [V+1,3] f32, voxel_tets [F,4] i64, vertex_code [V,3] f32 (the semantic code of the surface
    vertices, here the cmap)."""
    v = verts.reshape(-1, 3).astype(np.float64)
    centre = 0.5 * (v.min(0) + v.max(0))
    centre[1] = np.median(v[:, 1]) + 0.05
    vv = np.concatenate([v, centre[None]], 0) * 0.5
    vv[:, 2] *= -1.0
    f = faces.reshape(-1, 3).astype(np.int64)
    tets = np.concatenate([f, np.full((len(f), 1), len(v), np.int64)], 1)
    return vv.astype(np.float32), tets, cmap.reshape(-1, 3).astype(np.float32)


def stratified_points(verts: np.ndarray, faces: np.ndarray, n: int, seed: int = SEED) -> np.ndarray:
    """Query points [n,3] f32 mixing the regimes of SURVEY.md §7(1d): near-surface (face,
    edge and vertex Voronoi regions), inside/outside the clip band, far field and the cube
    boundary (|coord| >= 1)."""
    rng = np.random.RandomState(seed + 101)
    v = verts.astype(np.float64)
    fn = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-12)
    out = []
    q = n // 8
    # face interiors, small normal offsets (both sides)
    f = rng.randint(0, len(faces), q * 2)
    w = rng.dirichlet([1, 1, 1], q * 2)
    p = (v[faces[f]] * w[:, :, None]).sum(1)
    out.append(p + fn[f] * rng.uniform(-0.08, 0.08, (q * 2, 1)))
    # edge regions: points on an edge pushed along the averaged normal
    f = rng.randint(0, len(faces), q)
    t = rng.rand(q, 1)
    e = v[faces[f, 0]] * t + v[faces[f, 1]] * (1 - t)
    out.append(e + fn[f] * rng.uniform(0.0, 0.1, (q, 1)) + rng.normal(0, 0.002, (q, 3)))
    # vertex regions
    vi = rng.randint(0, len(v), q)
    vn = vertex_normals_f64(verts, faces)
    out.append(v[vi] + vn[vi] * rng.uniform(-0.03, 0.15, (q, 1)) + rng.normal(0, 0.002, (q, 3)))
    # uniform in the cube (far field + deep inside)
    out.append(rng.uniform(-1, 1, (q * 3, 3)))
    # on / outside the cube faces
    b = rng.uniform(-1.05, 1.05, (n - q * 7, 3))
    k = rng.randint(0, 3, len(b))
    b[np.arange(len(b)), k] = rng.choice([-1.0, 1.0, -1.02, 1.02], len(b))
    out.append(b)
    pts = np.concatenate(out, 0)
    rng.shuffle(pts)
    return pts[:n].astype(np.float32)


def lattice_points(res: int, z0: int = 0, z1: int | None = None) -> np.ndarray:
    """World coordinates of the marching-cubes lattice, float32 [N,3], ordered z slowest /
    x fastest - the mapping of lib/common/seg3d_lossless.py:125-137 with align_corners=True,
    b_min=[-1,1,-1], b_max=[1,-1,1] (apps/ICON.py:80-81): w = idx/(R-1) * (b_max-b_min) + b_min
    evaluated in float32 exactly as torch does."""
    z1 = res if z1 is None else z1
    idx = np.arange(res, dtype=np.float32) / np.float32(res - 1)
    xs = idx * np.float32(2.0) + np.float32(-1.0)
    ys = idx * np.float32(-2.0) + np.float32(1.0)
    zs = xs[z0:z1]
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
