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


# ---------------------------------------------------------------------------------------------
# the configurations outside configs/*.yaml that lib/net/MLP.py / HGPIFuNet.query also build: seeded constructions shared by
# tools/make_golden.py (the reference run on them -> tests/golden/variants.npz) and the GPU tests
# ---------------------------------------------------------------------------------------------
def weight_norm_state_dict(sd, seed=11):
    """norm_mlp 'weight' (lib/net/MLP.py:42-45): filters.l.weight_g / weight_v for all but the last layer, no norms.*"""
    rs = np.random.RandomState(seed)
    out = {}
    for k, v in sd.items():
        if k.startswith("norms."):
            continue
        if k.endswith(".weight") and not k.startswith("filters.3."):
            nrm = np.sqrt((v.reshape(len(v), -1) ** 2).sum(1)).reshape(-1, 1, 1).astype(np.float32)
            out[k + "_v"] = (v * rs.uniform(0.5, 2.0, (len(v), 1, 1))).astype(np.float32)      # any positive rescaling of v ...
            out[k + "_g"] = (nrm * rs.uniform(0.9, 1.1, nrm.shape)).astype(np.float32)         # ... is undone by g / ||v||
        else:
            out[k] = v
    return out


def callnorm_state_dict(sd, kind, seed=2):
    """norm_mlp 'group' (GroupNorm(32, C): affine) / 'instance' (InstanceNorm1d(C): no parameters), lib/net/MLP.py:35-41"""
    rs = np.random.RandomState(seed)
    out = {k: v for k, v in sd.items() if k.startswith("filters.")}
    if kind == "group":
        for l, c in enumerate((512, 256, 128)):
            out[f"norms.{l}.weight"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
            out[f"norms.{l}.bias"] = rs.normal(0, 0.1, c).astype(np.float32)
    return out


# name -> (feature planes used, smpl_feats, norm_mlp, last_op): what tests/golden/variants.npz holds the reference's answers for
VARIANTS = {
    "mvp_sdf": (6, ["sdf"], "batch", None),                            # configs/train/icon-mvp.yaml:40 (no 'vis': every plane is an input)
    "novis_full": (6, ["sdf", "norm", "cmap"], "batch", None),
    "vis_sdf": (12, ["sdf", "vis"], "batch", None),
    "weight": (12, ["sdf", "norm", "vis", "cmap"], "weight", None),
    "group": (12, ["sdf", "norm", "vis", "cmap"], "group", None),      # lib/common/config.py:80 - the config default
    "instance": (12, ["sdf", "norm", "vis", "cmap"], "instance", None),
    "sigmoid": (12, ["sdf", "norm", "vis", "cmap"], "batch", "sigmoid"),   # cfg.test_mode False (HGPIFuNet.py:133)
}


def variant_state_dict(name, a):
    planes, feats, norm, _ = VARIANTS[name]
    img = planes // 2 if "vis" in feats else planes
    c0 = img + 1 + (3 if "cmap" in feats else 0) + (3 if "norm" in feats else 0)
    sd = a.state_dict if c0 == 13 and len(feats) == 4 else synth.make_mlp_state_dict(synth.SEED + 5, dims=(c0, 512, 256, 128, 1))
    if norm == "weight":
        sd = weight_norm_state_dict(sd)
    elif norm in ("group", "instance"):
        sd = callnorm_state_dict(sd, norm)
    return c0, sd


def volume_encoder_replica(num_in=3, num_out=7, num_stacks=2):
    """A test-local module with EXACTLY the layer list of the reference's VolumeEncoder (lib/net/VE.py:114-183; Residual3D
    :55-111) - same attribute names, so the reference's state_dict loads with strict=True (tests/test_oracle_vs_reference.py
    checks that, and that both produce the same output) - for driving the real 128^3 -> 32^3 x 7 stack (k5 s2 d2 3-D
    convolutions, BatchNorm3d, two residual blocks with a dilated k3 convolution) through MIOpen on the GPU box, where
    /root/reference does not exist.  Call contract of HGPIFuNet.query: ve(vol, intermediate_output=False)[-1]."""
    import torch
    import torch.nn as nn

    class Residual3D(nn.Module):
        def __init__(self, n_in, n_out):
            super().__init__()
            self.numIn, self.numOut = n_in, n_out
            self.bn = nn.BatchNorm3d(n_in)                                   # constructed, never applied (VE.py:97-98)
            self.relu = nn.ReLU(inplace=True)
            self.conv1 = nn.Conv3d(n_in, n_out, kernel_size=3, stride=1, padding=2, dilation=2)
            self.bn1 = nn.BatchNorm3d(n_out)
            self.conv2 = nn.Conv3d(n_out, n_out, kernel_size=3, stride=1, padding=1)
            self.bn2 = nn.BatchNorm3d(n_out)
            self.conv3 = nn.Conv3d(n_out, n_out, kernel_size=3, stride=1, padding=1)      # constructed, never applied (:105-106)
            if n_in != n_out:
                self.conv4 = nn.Conv3d(n_in, n_out, kernel_size=1)

        def forward(self, x):
            out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
            return out + (self.conv4(x) if self.numIn != self.numOut else x)

    class VolumeEncoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.num_stacks = num_stacks
            self.relu = nn.ReLU(inplace=True)
            self.conv1 = nn.Conv3d(num_in, 8, kernel_size=5, stride=2, padding=4, dilation=2)
            self.bn1 = nn.BatchNorm3d(8)
            self.conv2 = nn.Conv3d(8, num_out, kernel_size=5, stride=2, padding=4, dilation=2)
            self.bn2 = nn.BatchNorm3d(num_out)
            self.conv_out1 = nn.Conv3d(num_out, num_out, kernel_size=3, stride=1, padding=1)   # constructed, never applied
            self.conv_out2 = nn.Conv3d(num_out, num_out, kernel_size=3, stride=1, padding=1)
            for idx in range(num_stacks):
                self.add_module("res" + str(idx), Residual3D(num_out, num_out))
            self.calls = 0

        def forward(self, x, intermediate_output=True):
            self.calls += 1
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.relu(self.bn2(self.conv2(out)))
            outs = []
            for idx in range(self.num_stacks):
                out = self._modules["res" + str(idx)](out)
                outs.append(out)
            return outs if intermediate_output else [outs[-1]]

    return VolumeEncoder()
