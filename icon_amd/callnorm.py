"""Regressors whose normalisation runs over the points of the call: ``norm_mlp: 'group'`` (``nn.GroupNorm(32, C)``, the
default of lib/common/config.py:80) and ``'instance'`` (``nn.InstanceNorm1d(C)``), lib/net/MLP.py:35-41,61-66.

Both normalise every hidden layer's output with the mean / biased variance over (the channels of a group) x (ALL N points
of the ``query()`` call), so a point's occupancy depends on which other points were asked for - the reference's adaptive
loop therefore evaluates each level with that level's own statistics, and a dense evaluation with the lattice's.  Given
the statistics the norm is a per-channel affine map, i.e. an eval-mode BatchNorm with ``running_mean / running_var`` set to
them, which is what the MLP kernels fold into their weights.  So a call is

  1. the MLP input rows of the call, materialised (``icon_query_rows`` / ``icon_grid_rows``, HIP);
  2. the statistics, layer by layer: plain f32 library GEMMs (``F.linear``, rocBLAS) over the rows in chunks - the layers below
     with their norms already folded in - per-channel sum and sum of squares of the layer's output accumulated in float64;
  3. ``icon_mlp_create`` with the statistics as BatchNorm operands and ``icon_mlp_forward`` on the rows (the f16x3 / f32 MFMA
     kernels), then the in_cube mask.

Cost: the MLP kernel plus about its own FLOPs again in f32 library GEMMs (2.3x when the pre-norm outputs of a layer do not
fit beside the rows and the layers below are recomputed) - this is the compatibility path of a configuration no
``configs/*.yaml`` ships (all use ``'batch'``), not the benchmarked one.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from ._lib import IconAmdError

CODE_IN_CUBE = 8            # csrc/common.h kCodeInCube: bit of the row's code word (slot 15) that says all(-1 < xyz < 1)
BN_EPS = 1e-5               # the eps icon_mlp_create folds with


class CallNormSpec:
    """what the norm layers of an ``MLP`` are: kind ('group' | 'instance'), groups per layer, eps per layer and the affine
    parameters per layer (None: InstanceNorm1d's default ``affine=False``)"""

    def __init__(self, kind: str, groups: Sequence[int], eps: Sequence[float], gamma: Sequence[Optional[torch.Tensor]],
                 beta: Sequence[Optional[torch.Tensor]]):
        self.kind, self.groups, self.eps, self.gamma, self.beta = kind, list(groups), list(eps), list(gamma), list(beta)


def spec_of(regressor, norm_mlp: Optional[str], widths: Sequence[int]) -> Optional[CallNormSpec]:
    """``regressor``: an ``MLP`` module (its ``norm`` / ``norms`` say what it is) or a state_dict (``norm_mlp`` must say it:
    a dict of a GroupNorm MLP has ``norms.l.weight / bias`` without running statistics, one of an InstanceNorm MLP no
    ``norms.*`` at all).  ``widths``: output channels of the hidden layers.  None: not a call-normalised regressor."""
    n = len(widths)
    if isinstance(regressor, dict):
        if norm_mlp not in ("group", "instance"):
            return None
        if norm_mlp == "group":
            for c in widths:
                if c % 32:
                    raise IconAmdError(f"GroupNorm(32, {c}): channels not divisible by the groups (lib/net/MLP.py:36)")
            gamma = [torch.as_tensor(regressor[f"norms.{l}.weight"]).detach().float().cpu() for l in range(n)]
            beta = [torch.as_tensor(regressor[f"norms.{l}.bias"]).detach().float().cpu() for l in range(n)]
            return CallNormSpec("group", [32] * n, [1e-5] * n, gamma, beta)
        has_affine = "norms.0.weight" in regressor
        gamma = [torch.as_tensor(regressor[f"norms.{l}.weight"]).detach().float().cpu() if has_affine else None for l in range(n)]
        beta = [torch.as_tensor(regressor[f"norms.{l}.bias"]).detach().float().cpu() if has_affine else None for l in range(n)]
        return CallNormSpec("instance", list(widths), [1e-5] * n, gamma, beta)
    kind = getattr(regressor, "norm", None)
    norms = list(getattr(regressor, "norms", ()))
    if kind not in ("group", "instance") or not norms:
        return None
    if len(norms) != n:
        raise IconAmdError(f"if_regressor has {len(norms)} norm layers for {n} hidden layers")
    groups, eps, gamma, beta = [], [], [], []
    for c, m in zip(widths, norms):
        if kind == "group":
            if not isinstance(m, torch.nn.GroupNorm):
                raise IconAmdError(f"if_regressor.norm = 'group' but norms hold {type(m).__name__}")
            groups.append(int(m.num_groups))
        else:
            if not isinstance(m, torch.nn.InstanceNorm1d):
                raise IconAmdError(f"if_regressor.norm = 'instance' but norms hold {type(m).__name__}")
            if getattr(m, "track_running_stats", False):
                raise IconAmdError("InstanceNorm1d(track_running_stats=True) is not what lib/net/MLP.py:39-41 builds")
            groups.append(int(c))
        eps.append(float(m.eps))
        w, b = getattr(m, "weight", None), getattr(m, "bias", None)
        gamma.append(w.detach().float().cpu() if w is not None else None)
        beta.append(b.detach().float().cpu() if b is not None else None)
    return CallNormSpec(kind, groups, eps, gamma, beta)


def _group_stats(mean_c: torch.Tensor, ey2_c: torch.Tensor, groups: int):
    """per-channel E[y], E[y^2] (float64, every channel over the same N points) -> per-channel mean / biased variance of the
    channel's group"""
    c = mean_c.numel()
    mg = mean_c.view(groups, c // groups).mean(1)
    eg = ey2_c.view(groups, c // groups).mean(1)
    var = (eg - mg * mg).clamp_min(0.0)
    rep = c // groups
    return mg.repeat_interleave(rep), var.repeat_interleave(rep)


@torch.no_grad()
def call_statistics(W: List[torch.Tensor], b: List[torch.Tensor], is_res: Sequence[bool], spec: CallNormSpec, rows: torch.Tensor,
                    c0: int, chunk: int = 1 << 20, keep_bytes: int = None):
    """mean / biased variance per channel (expanded from the groups) of every hidden layer's pre-norm output over the call:
    ``rows`` [N,16] device f32 (slots [0,c0) are the MLP input), ``W[l]`` [Cout,Cin] / ``b[l]`` device f32.  -> two lists of
    float64 device tensors."""
    dev = rows.device
    n_pts = rows.shape[0]
    n_hidden = len(W) - 1
    if n_pts == 0:
        raise IconAmdError("group / instance norm over an empty call")
    if keep_bytes is None:
        # the kept pre-norm outputs are an optimisation (they save recomputing the layers below): at most half of what is
        # free right now - a device with less head room recomputes instead of running out of memory
        keep_bytes = (torch.cuda.mem_get_info(dev)[0] // 2) if dev.type == "cuda" else (8 << 30)
    if is_res[0]:
        raise IconAmdError("the first layer cannot be a res layer")
    means, variances = [], []
    gam = [g.to(dev) if g is not None else None for g in spec.gamma]
    bet = [t.to(dev) if t is not None else None for t in spec.beta]

    def affine(k):
        """norm_k with the statistics found so far as y -> y * sc + sh (float64)"""
        sc = torch.rsqrt(variances[k] + spec.eps[k])
        if gam[k] is not None:
            sc = sc * gam[k].double()
        sh = -means[k] * sc
        if bet[k] is not None:
            sh = sh + bet[k].double()
        return sc, sh

    def folded(k):
        """layer k with its norm folded in, as eval-mode BatchNorm folds: one GEMM + LeakyReLU per layer instead of five
        elementwise passes over [chunk, C] (the fold moves the inputs of the next statistics by ~1e-7 relative)"""
        sc, sh = affine(k)
        return (W[k].double() * sc[:, None]).float(), (b[k].double() * sc + sh).float()

    # One pass over the rows per hidden layer.  The pre-norm outputs of the layer whose statistics were taken last are kept
    # for the next pass when they fit (257^3: 35 GB then 17 GB of the 288): the next layer is then one GEMM over them;
    # otherwise (513^3) the layers below are recomputed with their norms folded in.
    x_all = rows[:, :c0].contiguous()              # one strided copy instead of one per chunk and pass
    kept = None
    for l in range(n_hidden):
        cl = W[l].shape[0]
        s1 = torch.zeros(cl, dtype=torch.float64, device=dev)
        s2 = torch.zeros(cl, dtype=torch.float64, device=dev)
        below = [folded(k) for k in range(l)] if kept is None else None
        if kept is not None:
            kept_sc, kept_sh = (t.float() for t in affine(l - 1))
        keep = torch.empty((n_pts, cl), dtype=torch.float32, device=dev) if (l + 1 < n_hidden and n_pts * cl * 4 <= keep_bytes) else None
        for i in range(0, n_pts, chunk):
            x = x_all[i:i + chunk]
            if kept is not None:                       # y_{l-1} of this chunk is there: only its norm + activation remain
                h = F.leaky_relu_(torch.addcmul(kept_sh, kept[i:i + chunk], kept_sc), 0.01)
            else:
                h = x
                for k in range(l):
                    Wf, bf = below[k]
                    h = F.leaky_relu_(F.linear(torch.cat([h, x], 1) if is_res[k] else h, Wf, bf), 0.01)
            inp = torch.cat([h, x], 1) if is_res[l] else h
            # the GEMM writes straight into the kept buffer (a third of the path's GPU time went into copies before)
            y = torch.addmm(b[l], inp, W[l].t(), out=keep[i:i + chunk]) if keep is not None else F.linear(inp, W[l], b[l])
            s1 += y.sum(0, dtype=torch.float64)
            s2 += torch.linalg.vector_norm(y, dim=0, dtype=torch.float64) ** 2
        mu, var = _group_stats(s1 / n_pts, s2 / n_pts, spec.groups[l])
        means.append(mu)
        variances.append(var)
        kept = keep
    return means, variances


def batchnorm_equivalent(sd: dict, spec: CallNormSpec, means, variances) -> dict:
    """the regressor's state_dict with ``norms.l.*`` replaced by the eval-mode BatchNorm1d that computes the same map as the
    call's Group / InstanceNorm: running_mean / running_var = the call's statistics (variance shifted by the difference of
    the two layers' eps), weight / bias = the affine parameters (1 / 0 without)"""
    out = {k: v for k, v in sd.items() if not k.startswith("norms.")}
    for l, (mu, var) in enumerate(zip(means, variances)):
        c = mu.numel()
        out[f"norms.{l}.running_mean"] = mu.float().cpu()
        out[f"norms.{l}.running_var"] = (var + (spec.eps[l] - BN_EPS)).float().cpu()
        out[f"norms.{l}.weight"] = spec.gamma[l] if spec.gamma[l] is not None else torch.ones(c)
        out[f"norms.{l}.bias"] = spec.beta[l] if spec.beta[l] is not None else torch.zeros(c)
    return out


def in_cube_mask(rows: torch.Tensor) -> torch.Tensor:
    """the in_cube factor of HGPIFuNet.query (lib/net/HGPIFuNet.py:274-275,363) from the rows' code word"""
    return ((rows[:, 15].view(torch.int32) & CODE_IN_CUBE) != 0).to(torch.float32)
