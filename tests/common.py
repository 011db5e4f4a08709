"""Shared helpers for the test-suite (tests may use oracle/; the product never does)."""
import os
import sys
from functools import lru_cache

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from icon_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@lru_cache(maxsize=None)
def assets(mesh="body", prior="icon"):
    return synth.make_assets(mesh, prior_type=prior)


@lru_cache(maxsize=None)
def oracle_mlp(mesh="body", prior="icon"):
    return orc.Mlp(assets(mesh, prior).state_dict)


def oracle_query(a, pts, **kw):
    return orc.query_icon(a.smpl_verts[0], a.smpl_faces[0], a.smpl_cmap[0], a.smpl_vis[0], a.features,
                          orc.Mlp(a.state_dict), pts, sdf_clip=a.sdf_clip, **kw)


def vol_assets(prior):
    """Inputs of the pamir / pifu golden fixtures (tools/make_golden.py section e)."""
    feat = synth.make_feature_planes(6 if prior == "pamir" else 12, 128, synth.SEED)
    sd = synth.make_mlp_state_dict(synth.SEED + (1 if prior == "pamir" else 2), sdf_channel=None)
    vol = synth.make_feature_volume(7, 32, synth.SEED) if prior == "pamir" else None
    return feat, vol, sd


def rows16(x):
    """[N,c0] -> [N,16] zero padded point-major MLP input rows"""
    out = np.zeros((x.shape[0], 16), np.float32)
    out[:, : x.shape[1]] = x
    return out


def chamfer(va, fa, vb, fb, n=20000, seed=0):
    """Symmetric Chamfer distance between two triangle meshes, definition of
    lib/dataset/Evaluator.py:200-230 (mean closest-point distance both ways, average, x100),
    restated with area-weighted surface samples and the oracle's exact point-triangle distance."""
    def sample(v, f, rng):
        v = v.astype(np.float64)
        tri = v[f]
        area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
        idx = rng.choice(len(f), n, p=area / area.sum())
        r1, r2 = np.sqrt(rng.rand(n)), rng.rand(n)
        w = np.stack([1 - r1, r1 * (1 - r2), r1 * r2], 1)
        return (tri[idx] * w[:, :, None]).sum(1).astype(np.float32)
    rng = np.random.RandomState(seed)
    pa, pb = sample(va, fa, rng), sample(vb, fb, rng)
    d_ab = np.sqrt(orc.nearest_brute(vb.astype(np.float32), fb, pa)[0])
    d_ba = np.sqrt(orc.nearest_brute(va.astype(np.float32), fa, pb)[0])
    return 0.5 * (d_ab.mean() + d_ba.mean()) * 100.0, d_ab.mean() * 100.0


# what IconQueryEngine.attach() / query() read from the network object they replace the method of
# (lib/net/HGPIFuNet.py:48-165,236-245,268): tests/test_oracle_vs_reference.py checks the list against the reference's
# real module, tests/test_gpu_ties_shell.py drives the HIP path through a frozen replica carrying exactly these
ATTACH_NET_ATTRS = ("prior_type", "sdf_clip", "smpl_feats", "if_regressor", "smpl_feat_dict")
ATTACH_REGRESSOR_ATTRS = ("norm", "last_op", "res_layers", "norms", "filters", "training")
ATTACH_SMPL_KEYS = ("smpl_verts", "smpl_faces", "smpl_cmap", "smpl_vis")
