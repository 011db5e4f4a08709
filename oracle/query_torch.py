"""CPU port of the reference's query path with the SAME torch operators the reference runs
(F.grid_sample, torch.gather, Conv1d, BatchNorm1d, LeakyReLU, boolean-mask assignment), the three
third-party leaves coming from oracle/icon_oracle.c.  TEST INFRASTRUCTURE ONLY: it is the timed
`cpu_baseline` ("kind": "port") of bench.py - /root/reference does not exist on the GPU box - and a
second, independent restatement the CPU tests hold against the golden fixtures.

Follows, line by line in behaviour (not in text):
    lib/net/HGPIFuNet.py:268-367 (query), lib/dataset/mesh_util.py:266-277,319-396,
    lib/net/geometry.py:21-61, lib/net/MLP.py:8-72, lib/common/train_util.py:324-348.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import oracle as orc


class TorchMLP(nn.Module):
    """lib/net/MLP.py with norm='batch': filters / norms ModuleLists, same state_dict keys."""

    def __init__(self, dims=(13, 512, 256, 128, 1), res_layers=(2, 3, 4)):
        super().__init__()
        self.filters, self.norms, self.res_layers = nn.ModuleList(), nn.ModuleList(), tuple(res_layers)
        for l in range(len(dims) - 1):
            cin = dims[l] + (dims[0] if l in self.res_layers else 0)
            self.filters.append(nn.Conv1d(cin, dims[l + 1], 1))
            if l != len(dims) - 2:
                self.norms.append(nn.BatchNorm1d(dims[l + 1]))

    def forward(self, feature):
        y, x = feature, feature
        for i, f in enumerate(self.filters):
            y = f(y if i not in self.res_layers else torch.cat([y, x], 1))
            if i != len(self.filters) - 1:
                y = F.leaky_relu(self.norms[i](y), 0.01, inplace=True)     # the reference's nn.LeakyReLU(inplace=True), MLP.py:24
        return y


def build_mlp(state_dict) -> TorchMLP:
    m = TorchMLP()
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}, strict=False)
    return m.eval()


TIMES = {"leaves": 0.0}
_ACCEL = {}


def _accel(verts, faces):
    """one BVH + ray-bin structure per mesh (oracle/icon_accel.c): the exact accelerated leaves SURVEY.md
    section 8(d) asks the timed CPU path to use; bit-identical to the linear scans (tests/test_oracle_leaves.py)"""
    k = (verts.ctypes.data, faces.ctypes.data, verts.shape, faces.shape)
    if k not in _ACCEL:
        _ACCEL.clear()
        _ACCEL[k] = (orc.Accel(verts, faces), verts, faces)
    return _ACCEL[k][0]


def cal_sdf_batch(verts, faces, cmaps, vis, points):
    """mesh_util.py:357-396; tensors [1,V,3] [1,F,3] [1,V,3] [1,V,1] [1,N,3]"""
    import time
    t0 = time.perf_counter()
    acc = _accel(verts[0].numpy(), faces[0].numpy())
    leaf_sign = acc.check_sign(points[0].numpy())
    vn = torch.from_numpy(orc.vertex_normals(verts[0].numpy(), faces[0].numpy()))[None]
    fl = faces[0].long()
    tri, nrm, cm, vs = verts[0][fl], vn[0][fl], cmaps[0][fl], vis[0][fl]          # face_vertices
    d2, idx = acc.nearest(points[0].numpy())
    TIMES["leaves"] += time.perf_counter() - t0
    idx = torch.from_numpy(idx)
    ct, cn, cc, cv = tri[idx], nrm[idx], cm[idx], vs[idx]                          # gathers
    p = points[0]
    v0, v1, v2 = ct[:, 0], ct[:, 1], ct[:, 2]
    u, v = v1 - v0, v2 - v0
    n = torch.linalg.cross(u, v)
    s = (n * n).sum(1)
    s[s == 0] = 1e-6
    w = p - v0
    b2 = (torch.linalg.cross(u, w) * n).sum(1) / s
    b1 = (torch.linalg.cross(w, v) * n).sum(1) / s
    bw = torch.stack((1 - b1 - b2, b1, b2), -1)
    pts_cmap = (cc * bw[:, :, None]).sum(1)[None]
    pts_vis = (cv * bw[:, :, None]).sum(1)[None].ge(1e-1)
    pts_norm = (cn * bw[:, :, None]).sum(1)[None] * torch.tensor([-1.0, 1.0, -1.0])
    dist = torch.sqrt(torch.from_numpy(d2))[None] / torch.sqrt(torch.tensor(3.0))
    sign = 2.0 * (torch.from_numpy(leaf_sign)[None].float() - 0.5)
    return (dist * sign).unsqueeze(-1), pts_norm, pts_cmap, pts_vis


@torch.no_grad()
def query(assets, mlp: TorchMLP, points, sdf_clip=0.05):
    """HGPIFuNet.query, icon branch, identity calibration.  points [N,3] float32 numpy -> occ [N]"""
    T = torch.from_numpy
    xyz = T(np.ascontiguousarray(points.T))[None]                                   # [1,3,N]
    in_cube = ((xyz > -1.0) & (xyz < 1.0)).all(dim=1, keepdim=True).float()
    sdf, nrm, cm, vis = cal_sdf_batch(T(assets.smpl_verts), T(assets.smpl_faces), T(assets.smpl_cmap),
                                      T(assets.smpl_vis), xyz.permute(0, 2, 1).contiguous())
    outlier = torch.abs(sdf).ge(sdf_clip)
    sdf[outlier] = torch.sign(sdf[outlier])
    cm[outlier.repeat(1, 1, 3)] = sdf[outlier].repeat(1, 1, 3)                      # tiled list, as upstream
    smpl_feat = torch.cat([sdf, cm, nrm, vis.float()], dim=2).permute(0, 2, 1)
    feat = T(assets.features)
    samples = F.grid_sample(feat, xyz[:, :2].transpose(1, 2).unsqueeze(2), align_corners=True)[..., 0]
    dim = samples.shape[1] // 2
    sel = (1 - smpl_feat[:, [-1], :]).repeat(1, dim, 1) * dim + torch.arange(dim)[None, :, None]
    local = torch.gather(samples, 1, sel.long())
    pred = mlp(torch.cat([local, smpl_feat[:, :-1, :]], 1))
    return (in_cube * pred)[0, 0].numpy()
