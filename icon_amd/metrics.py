"""Mesh metrics of the reference's evaluator on the HIP closest-point engine.

``chamfer_p2s`` restates ``Evaluator.calculate_chamfer_p2s`` (lib/dataset/Evaluator.py:200-230):
surface samples on both meshes, closest-point distance to the other mesh both ways,
``chamfer = 0.5 * (mean d(gt samples -> pred mesh) + mean d(pred samples -> gt mesh)) * 100`` and
``p2s = mean d(gt samples -> pred mesh) * 100`` (the reference's variable naming: ``dist_pred_gt`` is
``closest_point(src_mesh = prediction, gt_surface_pts)``).  Differences, all on the safe side: the
reference draws 1,000 ``sample_surface_even`` points with trimesh (absent here); we draw ``n`` >= 100k
area-weighted uniform samples with a fixed seed, and the closest-point query is the exact
nearest-triangle kernel of this library (``icon_sdf_query``; |sdf| * sqrt(3) is the distance,
lib/dataset/mesh_util.py:391) instead of trimesh's rtree.

PyTorch is plumbing (sampling, reductions); the per-point work is the HIP kernel.
"""
from __future__ import annotations

import math

import torch

from .engine import MeshHandle
from ._lib import IconAmdError


def sample_surface(verts: torch.Tensor, faces: torch.Tensor, n: int, seed: int = 0) -> torch.Tensor:
    """[n,3] area-weighted uniform samples of a triangle mesh (device tensors)"""
    g = torch.Generator(device=verts.device).manual_seed(seed)
    tri = verts[faces.long()]                                        # [F,3,3]
    area = 0.5 * torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).norm(dim=1)
    idx = torch.multinomial(area / area.sum(), n, replacement=True, generator=g)
    r1 = torch.rand(n, device=verts.device, generator=g).sqrt()
    r2 = torch.rand(n, device=verts.device, generator=g)
    w = torch.stack([1 - r1, r1 * (1 - r2), r1 * r2], 1)
    return (tri[idx] * w[:, :, None]).sum(1).float().contiguous()


def _closest_distance(verts: torch.Tensor, faces: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    zeros3 = torch.zeros_like(verts)
    zeros1 = torch.zeros(verts.shape[0], device=verts.device)
    mesh = MeshHandle(verts, faces, zeros3, zeros1)
    out = mesh.sdf_query(pts)
    return out["sdf"].abs() * math.sqrt(3.0)


def chamfer_p2s(verts_pr, faces_pr, verts_gt, faces_gt, n: int = 100_000, seed: int = 0):
    """-> (chamfer, p2s), both x100 as in the reference.  Vertices in the units the caller wants the
    distances in (the reference evaluates in the [-1,1] cube: apps/ICON.py:758-759)."""
    if not verts_pr.is_cuda:
        raise IconAmdError("chamfer_p2s needs device tensors (there is no CPU path)")
    vp, vg = verts_pr.float().contiguous(), verts_gt.float().contiguous().to(verts_pr.device)
    fp, fg = faces_pr.long().contiguous(), faces_gt.long().contiguous().to(verts_pr.device)
    gt_pts = sample_surface(vg, fg, n, seed)
    pr_pts = sample_surface(vp, fp, n, seed + 1)
    d_pred_gt = _closest_distance(vp, fp, gt_pts)        # gt samples -> prediction mesh
    d_gt_pred = _closest_distance(vg, fg, pr_pts)        # prediction samples -> gt mesh
    d_pred_gt = torch.nan_to_num(d_pred_gt, nan=0.0)
    d_gt_pred = torch.nan_to_num(d_gt_pred, nan=0.0)
    chamfer = 0.5 * (d_pred_gt.mean() + d_gt_pred.mean()).item() * 100.0
    p2s = d_pred_gt.mean().item() * 100.0
    return chamfer, p2s


def to_unit_cube(verts: torch.Tensor, res: int) -> torch.Tensor:
    """apps/ICON.py:758-759: voxel units of export_mesh -> [-1,1]"""
    half = (res - 1) / 2.0
    return (verts - half) / half
