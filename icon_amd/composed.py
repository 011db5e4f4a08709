"""``HGPIFuNet.query`` for the regressors the fused kernels do not carry - composed from the HIP geometry leaf and
PyTorch-ROCm operators (lib/net/HGPIFuNet.py:268-367).

The kernels are built for what every ``configs/*.yaml`` ships: ``mlp_dim [*, 512, 256, 128, 1]``, ``res_layers [2, 3, 4]``, at most
15 MLP input channels, at most 16 feature channels per tap.  The reference's classes build more: PIFu-size regressors
(``[257, 1024, 512, 256, 128, 1]``), other ``res_layers``, 12 feature planes + all SMPL features without ``'vis'`` (19 inputs),
any ``last_op``.  For those the query is the reference's own sequence of operators with ONE substitution: ``cal_sdf_batch`` (kaolin
point_to_mesh_distance + check_sign + pytorch3d normals, lib/dataset/mesh_util.py:357-396) is ``icon_sdf_query`` - the packet-BVH
search, ray-bin inside test and barycentric attributes on the GPU.  Everything after it (clip, the tiled outlier-cmap rule,
``grid_sample``, ``feat_select``, the regressor MODULE itself, the in_cube mask) is PyTorch on the device, in chunks of 2 M points.

This is a compatibility path (one-time warning, ~10x the fused kernel's time), not the benchmarked one; it still needs the
HIP library and a GPU - there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from ._lib import IconAmdError

CHUNK = 1 << 21


def unsupported_reason(sd_shapes: Sequence[Sequence[int]], res_layers: Sequence[int], last_op_ok: bool, n_img: int, c0: int) -> Optional[str]:
    """why the fused kernels cannot evaluate this regressor / layout (None: they can).  ``sd_shapes``: [Cout, Cin] per layer"""
    couts = [int(s[0]) for s in sd_shapes]
    if couts != [512, 256, 128, 1]:
        return f"mlp_dim [*, {', '.join(str(c) for c in couts)}] (the kernels are built for [*, 512, 256, 128, 1])"
    res = sorted(l for l in res_layers if 0 < l < len(couts))
    if res != [2, 3]:
        return f"res_layers {list(res_layers)} (the kernels are built for [2, 3, 4])"
    if c0 > 15:
        return f"{c0} MLP input channels (the kernels carry 15)"
    if n_img > 16:
        return f"{n_img} feature channels per tap (the kernels carry 16)"
    if not last_op_ok:
        return "a last_op other than None / nn.Sigmoid"
    return None


class FunctionalMLP:
    """``MLP.forward`` (lib/net/MLP.py:49-72) from a state_dict, for regressors handed over as one: Conv1d(k=1) stack, norm per
    ``norm_mlp`` ('batch': eval-mode running statistics; 'group': GroupNorm(32); 'instance'; anything else / no ``norms.*``: none),
    LeakyReLU(0.01), input re-concatenated before the layers in ``res_layers``, ``last_op`` None or 'sigmoid'"""

    def __init__(self, sd: dict, res_layers, norm_mlp: Optional[str], last_op: Optional[str], device):
        n = 0
        while f"filters.{n}.weight" in sd:
            n += 1
        if n == 0:
            raise IconAmdError("state_dict has no filters.0.weight")
        t = lambda v: torch.as_tensor(v).detach().to(device, torch.float32)
        self.W = [t(sd[f"filters.{l}.weight"]).reshape(sd[f"filters.{l}.weight"].shape[0], -1, 1) for l in range(n)]
        self.b = [t(sd[f"filters.{l}.bias"]) for l in range(n)]
        if last_op not in (None, "sigmoid"):
            raise IconAmdError(f"last_op {last_op!r}: a state_dict regressor can carry None or 'sigmoid' (pass the module for anything else)")
        self.res_layers, self.last_op = tuple(res_layers), last_op
        if "norms.0.running_mean" in sd:
            self.kind = "batch"
        elif norm_mlp in ("group", "instance"):
            self.kind = norm_mlp
        else:
            self.kind = None
        self.norm = []
        for l in range(n - 1):
            g = {k: t(sd[f"norms.{l}.{k}"]) for k in ("weight", "bias", "running_mean", "running_var") if f"norms.{l}.{k}" in sd}
            self.norm.append(g)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        y, x0 = x, x
        n = len(self.W)
        for l in range(n):
            y = F.conv1d(torch.cat([y, x0], 1) if l in self.res_layers else y, self.W[l], self.b[l])
            if l != n - 1:
                g = self.norm[l]
                if self.kind == "batch":
                    y = F.batch_norm(y, g["running_mean"], g["running_var"], g.get("weight"), g.get("bias"), False, 0.0, 1e-5)
                elif self.kind == "group":
                    y = F.group_norm(y, 32, g.get("weight"), g.get("bias"), 1e-5)
                elif self.kind == "instance":
                    y = F.instance_norm(y, None, None, g.get("weight"), g.get("bias"), True, 0.0, 1e-5)
                y = F.leaky_relu(y, 0.01)
        if self.last_op == "sigmoid":
            y = torch.sigmoid(y)
        return y


def tiled_outlier_cmap(cmap: torch.Tensor, sdf: torch.Tensor, outlier: torch.Tensor, local: bool) -> torch.Tensor:
    """lib/net/HGPIFuNet.py:303-305 - ``smpl_cmap[outlier.repeat(1,1,3)] = smpl_sdf[outlier].repeat(1,1,3)``: the K outlier signs,
    tiled three times, are consumed in row-major order, so outlier j's channel k receives the sign of outlier (3j+k) mod K
    (``local``: the evidently intended rule - its own sign)"""
    s = sdf[outlier]                                    # [K] in point order
    k = s.numel()
    if k == 0:
        return cmap
    out = cmap.clone()
    if local:
        out[outlier.expand(-1, -1, 3)] = s.repeat_interleave(3)
    else:
        idx = (3 * torch.arange(k, device=s.device)[:, None] + torch.arange(3, device=s.device)[None, :]) % k
        out[outlier.expand(-1, -1, 3)] = s[idx].reshape(-1)
    return out


@torch.no_grad()
def query_composed(eng, features, points: torch.Tensor, calibs: torch.Tensor, regressor, sdf_query=None):
    """features: list of [1,C,H,W]; points [1,3,N]; calibs [1,4,4] (device) -> list of [1,1,N].  ``regressor``: a module (called as
    it is) or a state_dict (FunctionalMLP).  ``sdf_query``: the geometry leaf, points [N,3] -> dict (default: the engine's mesh
    handle - icon_sdf_query; the CPU tests inject the checker's)."""
    dev = points.device
    n = int(points.shape[2])
    calibs = calibs.to(dev, torch.float32)
    xyz = torch.baddbmm(calibs[:, :3, 3:4], calibs[:, :3, :3], points.to(torch.float32))           # orthogonal(), geometry.py:54-56
    in_cube = ((xyz > -1.0) & (xyz < 1.0)).all(dim=1, keepdim=True).float()                         # HGPIFuNet.py:274-275
    feats = tuple(getattr(eng, "smpl_feats", ("sdf", "norm", "vis", "cmap")))
    smpl_feat = None
    if eng.prior_type == "icon":
        if sdf_query is None:
            sdf_query = eng._mesh_handle().sdf_query
        o = sdf_query(xyz[0].t().contiguous())
        sdf = o["sdf"].reshape(1, n, 1).to(torch.float32).clone()
        norm, cmap = o["norm"].reshape(1, n, 3).to(torch.float32), o["cmap"].reshape(1, n, 3).to(torch.float32)
        vis = o["vis"].reshape(1, n, 1).to(torch.float32)
        outlier = sdf.abs() >= eng.sdf_clip                                                         # :298-299
        sdf[outlier] = torch.sign(sdf[outlier])
        lst = [sdf]
        if "cmap" in feats:
            lst.append(tiled_outlier_cmap(cmap, sdf, outlier, eng.cmap_mode == "local"))
        if "norm" in feats:
            lst.append(norm)
        if "vis" in feats:
            lst.append(vis)
        smpl_feat = torch.cat(lst, 2).permute(0, 2, 1)                                              # [1, c, N]
    vol = eng._pamir_volume() if eng.prior_type == "pamir" else None
    if isinstance(regressor, dict):
        from .engine import effective_filters
        regressor = FunctionalMLP(effective_filters(regressor), eng.res_layers, eng.norm_mlp, eng.last_op, dev)
        per_call_norm = regressor.kind in ("group", "instance")
    else:
        per_call_norm = getattr(regressor, "norm", None) in ("group", "instance") and len(getattr(regressor, "norms", ())) > 0
    step = max(n, 1) if per_call_norm else CHUNK  # Group / InstanceNorm: the statistics are the call's - no chunking (an empty call: no pass)
    preds = []
    for im_feat in features:
        im_feat = im_feat.to(torch.float32)
        out = torch.empty((1, 1, n), dtype=torch.float32, device=dev)
        for a in range(0, n, step):
            b = min(n, a + step)
            uv = xyz[:, :2, a:b].transpose(1, 2).unsqueeze(2)                                       # index(), geometry.py:21-43
            local = F.grid_sample(im_feat, uv, align_corners=True)[:, :, :, 0]
            if eng.prior_type == "icon":
                sf = smpl_feat[:, :, a:b]
                if "vis" in feats:                                                                   # feat_select, mesh_util.py:266-277
                    half = local.shape[1] // 2
                    idx = ((1.0 - sf[:, -1:, :]) * half + torch.arange(half, device=dev).view(1, half, 1)).long()
                    point_feat = torch.cat([torch.gather(local, 1, idx), sf[:, :-1, :]], 1)
                else:
                    point_feat = torch.cat([local, sf], 1)
            elif eng.prior_type == "pamir":
                g3 = xyz[:, :, a:b].transpose(1, 2)[:, :, None, None, :]
                point_feat = torch.cat([local, F.grid_sample(vol, g3, align_corners=True)[:, :, :, 0, 0]], 1)
            else:
                point_feat = torch.cat([local, xyz[:, 2:3, a:b]], 1)
            prev = torch.backends.cudnn.enabled                  # MIOpen's batch norm refuses [1, C, 2 M] (miopenStatusBadParm): ATen's own kernels
            torch.backends.cudnn.enabled = False
            try:
                out[:, :, a:b] = in_cube[:, :, a:b] * regressor(point_feat)                         # :361-363
            finally:
                torch.backends.cudnn.enabled = prev
        preds.append(out)
    return preds
