"""Host-side mirror of the reference's query interface, on top of the C ABI (include/icon_amd.h).

What a user of the reference sees (paths relative to the reference root):

* ``IconQueryEngine.query(features, points, calibs, transforms=None, regressor=None)``
  == ``HGPIFuNet.query`` (lib/net/HGPIFuNet.py:268-367): same arguments, same ``[ [1,1,N] ]`` return.
* ``query_func(opt, netG, features, points, proj_matrix=None)``
  == lib/common/train_util.py:324-348.
* ``DenseReconEngine`` == the ``reconEngine`` object (``Seg3dLossless``, lib/common/seg3d_lossless.py:36,
  constructed at apps/ICON.py:78-90): ``forward(**kwargs) -> Tensor[D,H,W] | None``, ``export_mesh``,
  ``resolutions`` / ``b_min`` / ``b_max`` buffers; it is an ``nn.Module`` so the checkpoint filters
  that look for the substring ``reconEngine`` keep working (lib/dataset/mesh_util.py:203).

PyTorch is plumbing here (device memory, streams, torch.distributed); all per-point arithmetic is
in the HIP kernels.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import IconAmdError, check, ptr


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on(t: torch.Tensor):
    """Make the device of ``t`` current for the duration of a library call: the C ABI allocates and launches on HIP's
    current device, and one process may drive several (cfg 5: one image per GPU)."""
    return torch.cuda.device(t.device) if t.is_cuda else contextlib.nullcontext()


def _key(*tensors) -> tuple:
    """Identity of cached source tensors.  A key is only trusted together with STRONG references to
    the tensors it was made from (kept next to every cached handle): as long as those are alive the
    caching allocator cannot hand the same address to the next image's tensors, so an equal
    (data_ptr, _version, shape) really is the same data."""
    return tuple((t.data_ptr(), t._version, t.shape, t.device) for t in tensors)      # (torch.Size / torch.device compare by value)


def _dev_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise IconAmdError(f"{what} is on {t.device}; icon_amd has no CPU path - move it to the HIP device")
    return t.detach().to(torch.float32).contiguous()


class _Handle:
    _destroy = None

    def __init__(self):
        self.h = C.c_void_p(0)

    def close(self):
        if self.h and self.h.value:
            getattr(_lib.lib(), self._destroy)(self.h)
            self.h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MeshHandle(_Handle):
    """Per-image SMPL body (icon_mesh_create_arena): the tensors of ``smpl_feat_dict``
    (lib/net/HGPIFuNet.py:236-240), batch size 1.  Built by kernels on the current stream into a block of the caching
    allocator: no copy of the mesh to the host, no allocation in the steady state, no synchronisation - as the reference's
    own per-call prologue runs on the device (lib/dataset/mesh_util.py:367-372).  ``validate=True`` (the default for a
    handle made by hand) waits for the build and raises on bad input as the round-3 host build did; the engine passes
    False and polls ``status()`` on later calls instead (an error surfaces one call late, nothing ever faults)."""
    _destroy = "icon_mesh_destroy"

    def __init__(self, smpl_verts, smpl_faces, smpl_cmap, smpl_vis, validate: bool = True):
        super().__init__()
        _lib.require_device()
        v = _dev_f32(smpl_verts, "smpl_verts").reshape(-1, 3)
        f = smpl_faces.detach()
        if not f.is_cuda:
            raise IconAmdError("smpl_faces must be on the HIP device")
        f = f.to(torch.int64).reshape(-1, 3).contiguous()
        cm = _dev_f32(smpl_cmap, "smpl_cmap").reshape(-1, 3)
        vs = _dev_f32(smpl_vis, "smpl_vis").reshape(-1)
        if cm.shape[0] != v.shape[0] or vs.shape[0] != v.shape[0]:
            raise IconAmdError("smpl_cmap / smpl_vis do not match smpl_verts")
        self.V, self.F = int(v.shape[0]), int(f.shape[0])
        self.device = v.device
        self.checked = False
        nbytes = C.c_int64(0)
        check(_lib.lib().icon_mesh_arena_bytes(C.c_int64(self.V), C.c_int64(self.F), C.byref(nbytes)), "icon_mesh_arena_bytes")
        with _on(v):
            # every array of the mesh lives in this block (freed - to the allocator's cache, in stream order - with the handle)
            self.arena = torch.empty(int(nbytes.value), dtype=torch.uint8, device=v.device)
            check(_lib.lib().icon_mesh_create_arena(ptr(v), C.c_int64(self.V), ptr(f), C.c_int64(self.F), ptr(cm), ptr(vs),
                                                    ptr(self.arena), nbytes, _stream(), C.byref(self.h)), "icon_mesh_create")
        if validate:
            self.status(wait=True)

    def status(self, wait: bool = False) -> Optional[int]:
        """Input check of the device build (icon_mesh_status): raises IconAmdError for a face naming a missing vertex or a
        non-finite coordinate; returns the status bits, or None while the build has not run yet (``wait=False`` never blocks)."""
        if self.checked:
            return 0
        bits = C.c_int(0)
        check(_lib.lib().icon_mesh_status(self.h, C.c_int(int(wait)), C.byref(bits)), "icon_mesh_create")
        if bits.value < 0:
            return None
        self.checked = True
        return int(bits.value)

    def vertex_normals(self) -> torch.Tensor:
        out = torch.empty((self.V, 3), dtype=torch.float32, device=self.device)
        check(_lib.lib().icon_mesh_vertex_normals(self.h, ptr(out), _stream()), "icon_mesh_vertex_normals")
        return out

    def stats(self) -> dict:
        arr = (C.c_int64 * 6)()
        check(_lib.lib().icon_mesh_stats(self.h, arr), "icon_mesh_stats")
        return dict(nodes=arr[0], depth=arr[1], bin_entries=arr[2], max_bin=arr[3], leaves=arr[4], slots=arr[5])

    def traversal_stats(self, res: int, z0: int = 0, z1: Optional[int] = None) -> dict:
        arr = (C.c_uint64 * 4)()
        check(_lib.lib().icon_debug_traversal_stats(self.h, C.c_int(res), C.c_int(z0), C.c_int(res if z1 is None else z1), arr),
              "icon_debug_traversal_stats")
        return dict(waves=arr[0], nodes=arr[1], tris=arr[2], nodes_per_wave=arr[1] / max(arr[0], 1),
                    tris_per_wave=arr[2] / max(arr[0], 1), longest_walk=arr[3])

    def sdf_query(self, points: torch.Tensor, search: str = "bvh"):
        """cal_sdf_batch (lib/dataset/mesh_util.py:357-396) for points [N,3] ->
        dict(sdf [N], norm [N,3], cmap [N,3], vis [N], face [N] i64, inside [N] bool)"""
        p = _dev_f32(points, "points").reshape(-1, 3)
        n = p.shape[0]
        dev = p.device
        out = dict(sdf=torch.empty(n, device=dev), norm=torch.empty((n, 3), device=dev),
                   cmap=torch.empty((n, 3), device=dev), vis=torch.empty(n, device=dev),
                   face=torch.empty(n, dtype=torch.int64, device=dev),
                   inside=torch.empty(n, dtype=torch.uint8, device=dev))
        check(_lib.lib().icon_sdf_query(self.h, ptr(p), C.c_int64(n), ptr(out["sdf"]), ptr(out["norm"]),
                                        ptr(out["cmap"]), ptr(out["vis"]), ptr(out["face"]), ptr(out["inside"]),
                                        C.c_int(_lib.SEARCH[search]), _stream()), "icon_sdf_query")
        out["inside"] = out["inside"].bool()
        return out


    def sdf_query_ties(self, points: torch.Tensor) -> dict:
        """Tie sensitivity of the nearest-triangle choice (lib/dataset/mesh_util.py:374-390) for points [N,3] ->
        dict(face [N] i32 the winner, face2 [N] i32 the runner-up (-1: none), ulps [N] u8: float32 ulps between their
        squared distances, clipped to 255).  ulps <= 1 marks a point whose norm / cmap / vis depend on how the leaf
        rounds its last bit - the part of the output that is not comparable with a kaolin run point by point."""
        p = _dev_f32(points, "points").reshape(-1, 3)
        n, dev = p.shape[0], p.device
        out = dict(face=torch.empty(n, dtype=torch.int32, device=dev), face2=torch.empty(n, dtype=torch.int32, device=dev),
                   ulps=torch.empty(n, dtype=torch.uint8, device=dev))
        check(_lib.lib().icon_sdf_query_ties(self.h, ptr(p), C.c_int64(n), ptr(out["face"]), ptr(out["face2"]), ptr(out["ulps"]),
                                             _stream()), "icon_sdf_query_ties")
        return out


class FeatHandle(_Handle):
    """Feature planes ``features[-1]`` of HGPIFuNet.filter ([1,C,H,W]) and, for PaMIR, the volume
    encoder output ([1,Cv,D,H,W]); icon_feat_create."""
    _destroy = "icon_feat_destroy"

    def __init__(self, planes: torch.Tensor, n_select: int, vol: Optional[torch.Tensor] = None,
                 smpl_feats: Sequence[str] = ("sdf", "norm", "vis", "cmap")):
        super().__init__()
        _lib.require_device()
        p = _dev_f32(planes, "features")
        if p.dim() == 4:
            if p.shape[0] != 1:
                raise IconAmdError("batch size must be 1 (lib/common/seg3d_lossless.py:73)")
            p = p[0]
        p = p.contiguous()
        Cc, H, W = (int(s) for s in p.shape)
        vp, Cv, Dv, Hv, Wv = C.c_void_p(0), 0, 0, 0, 0
        if vol is not None:
            v = _dev_f32(vol, "vol_feat")
            if v.dim() == 5:
                v = v[0]
            v = v.contiguous()
            Cv, Dv, Hv, Wv = (int(s) for s in v.shape)
            vp = ptr(v)
        self.C, self.H, self.W, self.n_select, self.Cv = Cc, H, W, n_select, Cv
        with _on(p):
            check(_lib.lib().icon_feat_create(ptr(p), C.c_int(Cc), C.c_int(H), C.c_int(W), C.c_int(n_select), vp,
                                              C.c_int(Cv), C.c_int(Dv), C.c_int(Hv), C.c_int(Wv), _stream(),
                                              C.byref(self.h)), "icon_feat_create")
        if set(smpl_feats) | {"vis"} != {"sdf", "norm", "vis", "cmap"}:
            check(_lib.lib().icon_feat_set_smpl_feats(self.h, C.c_int(int("cmap" in smpl_feats)), C.c_int(int("norm" in smpl_feats))),
                  "icon_feat_set_smpl_feats")
        # no synchronisation: the repack kernel is enqueued on the current stream, and the caching allocator
        # recycles the (possibly temporary) source tensors in stream order


def _np32(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        t = t.detach().to("cpu", torch.float32).numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def effective_filters(state_dict: dict) -> dict:
    """``norm_mlp: 'weight'`` (lib/net/MLP.py:42-45): ``nn.utils.weight_norm`` keeps ``filters.l.weight_g`` / ``weight_v`` and
    applies ``v * (g / ||v||)`` (norm over all but the output dimension) - returned as ``filters.l.weight``.  Other
    state_dicts pass through."""
    if not any(k.endswith("weight_g") for k in state_dict):
        return state_dict
    out = {k: v for k, v in state_dict.items() if not (k.endswith("weight_g") or k.endswith("weight_v"))}
    for k, g in state_dict.items():
        if k.endswith("weight_g"):
            v = state_dict[k[:-1] + "v"]
            g, v = (torch.as_tensor(t).detach().float().cpu() for t in (g, v))
            out[k[:-2]] = torch._weight_norm(v, g, 0)
    return out


class MlpHandle(_Handle):
    """``if_regressor`` weights in reference state_dict layout (``filters.{l}.weight [Cout,Cin,1]``,
    ``filters.{l}.bias``, ``norms.{l}.{weight,bias,running_mean,running_var}``; lib/net/MLP.py:26-45);
    icon_mlp_create folds BatchNorm and packs the MFMA operands."""
    _destroy = "icon_mlp_destroy"

    def __init__(self, state_dict, res_layers: Sequence[int] = (2, 3, 4), bn_eps: float = 1e-5, last_op: Optional[str] = None):
        super().__init__()
        if last_op not in (None, "sigmoid"):
            raise IconAmdError(f"unsupported last_op {last_op!r} (None or 'sigmoid')")
        _lib.require_device()
        state_dict = effective_filters(state_dict)
        n = 0
        while f"filters.{n}.weight" in state_dict:
            n += 1
        if n == 0:
            raise IconAmdError("state_dict has no filters.0.weight")
        W = [_np32(state_dict[f"filters.{l}.weight"]) for l in range(n)]
        W = [w.reshape(w.shape[0], -1) for w in W]
        b = [_np32(state_dict[f"filters.{l}.bias"]) for l in range(n)]
        has_bn = f"norms.0.running_mean" in state_dict
        cin = (C.c_int * n)(*[w.shape[1] for w in W])
        cout = (C.c_int * n)(*[w.shape[0] for w in W])
        is_res = (C.c_int * n)(*[1 if l in tuple(res_layers) else 0 for l in range(n)])
        self.c0 = int(W[0].shape[1])

        def parr(arrs):
            return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])

        keep = [W, b]
        if has_bn:
            bn = [[_np32(state_dict[f"norms.{l}.{k}"]) for l in range(n - 1)]
                  for k in ("weight", "bias", "running_mean", "running_var")]
            keep.append(bn)
            bn_ptrs = [parr(x) for x in bn]
        else:
            bn_ptrs = [None] * 4
        check(_lib.lib().icon_mlp_create(C.c_int(n), cin, cout, is_res, parr(W), parr(b), bn_ptrs[0], bn_ptrs[1],
                                         bn_ptrs[2], bn_ptrs[3], C.c_float(bn_eps), _stream(), C.byref(self.h)),
              "icon_mlp_create")
        self.last_op = last_op
        if last_op == "sigmoid":
            check(_lib.lib().icon_mlp_set_last_op(self.h, C.c_int(1)), "icon_mlp_set_last_op")

    def forward(self, x: torch.Tensor, precision: str = "f16x3") -> torch.Tensor:
        """MLP.forward on point-major rows x [N,16] (slots >= c0 ignored) -> [N]"""
        x = _dev_f32(x, "x")
        if x.dim() != 2 or x.shape[1] != 16:
            raise IconAmdError("x must be [N,16]")
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        check(_lib.lib().icon_mlp_forward(self.h, ptr(x), C.c_int64(x.shape[0]), ptr(out),
                                          C.c_int(_lib.PRECISION[precision]), _stream()), "icon_mlp_forward")
        return out



_WARNED_VOXELIZER = False
_WARNED_COMPOSED = set()      # reasons the composed path was announced for


class Workspace(_Handle):
    _destroy = "icon_work_destroy"

    def __init__(self):
        super().__init__()
        check(_lib.lib().icon_work_create(C.byref(self.h)), "icon_work_create")

    def set_tie_rule(self, rule: int, ulps: int = 0) -> None:
        check(_lib.lib().icon_work_set_tie_rule(self.h, C.c_int(rule), C.c_int(ulps)), "icon_work_set_tie_rule")

    def status(self) -> None:
        """Raise IconAmdError if a shared-walk search launched on this workspace gave up on a hand-over (icon_work_status: a
        host read of the workspace's error record, no synchronisation - synchronise first for the verdict on a given launch)."""
        check(_lib.lib().icon_work_status(self.h), "icon_work_status")

    def set_reserve_cus(self, n: int) -> None:
        """leave ``n`` CUs free of the persistent MLP kernel (for the RCCL kernels of an overlapped all_gather)"""
        check(_lib.lib().icon_work_set_reserve_cus(self.h, C.c_int(int(n))), "icon_work_set_reserve_cus")

    def profile(self, enable: bool = True) -> None:
        check(_lib.lib().icon_work_profile(self.h, C.c_int(int(enable))), "icon_work_profile")

    def profile_detail(self) -> dict:
        """of the most recent profiled call (icon_work_profile_detail): the search kernel alone, and the shader cycles / wall time
        / effective clock of the fused MLP kernel's workgroup 0"""
        out = (C.c_double * 4)()
        check(_lib.lib().icon_work_profile_detail(self.h, out), "icon_work_profile_detail")
        return {"nearest_ms": float(out[0]), "fused_cycles": float(out[1]), "fused_wall_ms": float(out[2]), "effective_clock_mhz": float(out[3])}

    def profile_workgroups(self):
        """of the most recent profiled call (icon_work_profile_workgroups): one row per workgroup of the fused MLP kernel's
        persistent grid - XCD id, start (ms after the first start), span (ms), shader cycles, tiles evaluated"""
        import numpy as np
        cap = 1024
        out = (C.c_double * (5 * cap))()
        n = C.c_int(0)
        check(_lib.lib().icon_work_profile_workgroups(self.h, out, C.c_int(cap), C.byref(n)), "icon_work_profile_workgroups")
        return np.frombuffer(out, dtype=np.float64)[:5 * min(n.value, cap)].reshape(-1, 5).copy()

    def set_steal(self, permille: int, group: int = 2) -> None:
        """the fused MLP kernel's tile partition: the last ``permille``/1000 of a launch's tiles are drawn dynamically in
        contiguous groups of ``group`` tiles (0 = all static); the result does not depend on it"""
        check(_lib.lib().icon_work_set_steal(self.h, C.c_int(int(permille)), C.c_int(int(group))), "icon_work_set_steal")

    def adaptive_reruns(self) -> int:
        """schedules (icon_adaptive_eval) on this workspace that were run a second time with the per-launch range rescue"""
        n = C.c_int(0)
        check(_lib.lib().icon_adaptive_reruns(self.h, C.byref(n)), "icon_adaptive_reruns")
        return int(n.value)

    def stage_ms(self):
        """(features_ms, patch_ms, mlp_ms) of the most recent call, from HIP events on its stream"""
        out = (C.c_float * 3)()
        check(_lib.lib().icon_work_stage_ms(self.h, out), "icon_work_stage_ms")
        return float(out[0]), float(out[1]), float(out[2])


def check_regressor(regressor, norm_mlp: Optional[str] = None) -> None:
    """Refuse every ``MLP`` configuration the kernels do not evaluate (lib/net/MLP.py:8-72): eval-mode BatchNorm1d
    (``norm_mlp: 'batch'`` in every configs/*.yaml) folds into the weights, ``'weight'`` (weight_norm, no norm layers) and any
    other string (no norm at all, MLP.py:64-65) are plain layers; the config default 'group', lib/common/config.py:80, and
    'instance' normalise over the points of the call and take the per-call path of icon_amd/callnorm.py (a state_dict cannot
    say which it is: ``norm_mlp`` must).  ``last_op`` may be None (``test_mode: True``) or ``nn.Sigmoid``
    (lib/net/HGPIFuNet.py:128-133)."""
    if isinstance(regressor, dict):
        if any(k.startswith("norms.") for k in regressor) and "norms.0.running_mean" not in regressor and norm_mlp not in ("group", "instance"):
            raise IconAmdError("regressor state_dict has norms.* without running statistics (GroupNorm / affine InstanceNorm): set "
                               "engine.norm_mlp = 'group' or 'instance' - a state_dict does not say which")
        return
    norm = getattr(regressor, "norm", "batch")
    has_norm_layers = len(getattr(regressor, "norms", ())) > 0
    if has_norm_layers and norm not in ("batch", "group", "instance"):
        raise IconAmdError(f"if_regressor.norm = {norm!r} with norm layers: lib/net/MLP.py builds norm layers for 'batch', "
                           "'group' and 'instance' only")
    lo = getattr(regressor, "last_op", None)
    if lo is not None and not isinstance(lo, nn.Sigmoid):
        raise IconAmdError(f"if_regressor.last_op = {type(lo).__name__}: only None (cfg.test_mode) and nn.Sigmoid "
                           "(lib/net/HGPIFuNet.py:133) are evaluated")
    if getattr(regressor, "training", False) and norm == "batch":
        raise IconAmdError("if_regressor is in training mode: BatchNorm batch statistics cannot be folded - call .eval()")


def regressor_last_op(regressor) -> Optional[str]:
    """'sigmoid' for a module built with last_op=nn.Sigmoid() (cfg.test_mode False), else None; dicts carry no last_op"""
    return "sigmoid" if isinstance(getattr(regressor, "last_op", None), nn.Sigmoid) else None


def regressor_state_dict(regressor, norm_mlp: Optional[str] = None) -> dict:
    """state_dict of an ``MLP`` module (lib/net/MLP.py) or a dict already in that layout."""
    check_regressor(regressor, norm_mlp)
    if isinstance(regressor, dict):
        return regressor
    return {k: v for k, v in regressor.state_dict().items() if "num_batches_tracked" not in k}


def _device_of(*cands):
    for c in cands:
        if isinstance(c, (list, tuple)) and c:
            c = c[-1]
        if isinstance(c, torch.Tensor) and c.is_cuda:
            return c
    return None


def _guarded(fn):
    """run the method with the device of its first device-tensor argument current (see _on)"""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        t = _device_of(*args, *kw.values())
        if t is None:
            return fn(self, *args, **kw)
        with _on(t):
            return fn(self, *args, **kw)
    return wrapper


class IconQueryEngine:
    """Drop-in for ``HGPIFuNet.query`` (lib/net/HGPIFuNet.py:268-367).

    Either attach it to a constructed reference network (``IconQueryEngine.attach(netG)`` replaces
    ``netG.query`` and reads ``netG.smpl_feat_dict / sdf_clip / prior_type / smpl_feats`` exactly
    as the method does), or use it standalone via ``set_mesh`` / ``set_regressor``.
    """

    def __init__(self, prior_type: str = "icon", sdf_clip: float = 0.05,
                 smpl_feats: Sequence[str] = ("sdf", "norm", "vis", "cmap"),
                 cmap_mode: str = "reference", search: str = "bvh", precision: str = "f16x3",
                 res_layers: Sequence[int] = (2, 3, 4), voxelizer: str = "auto"):
        if prior_type not in _lib.PRIOR:
            raise IconAmdError(f"unknown prior_type {prior_type!r}")
        if prior_type == "icon":
            unknown = set(smpl_feats) - {"sdf", "norm", "vis", "cmap"}
            if unknown:
                raise IconAmdError(f"smpl_feats = {list(smpl_feats)}: only sdf / norm / vis / cmap exist")
            # the sdf is the first smpl feature whether or not it is listed (lib/net/HGPIFuNet.py:300)
            # without 'vis' (configs/train/icon-mvp.yaml:40) the reference feeds EVERY feature channel, not the half smpl_vis
            # selects (lib/net/HGPIFuNet.py:345-346); the kernels carry at most 15 input channels in all and refuse more
        self.smpl_feats = tuple(smpl_feats)
        self.prior_type, self.sdf_clip = prior_type, float(sdf_clip)
        self.cmap_mode, self.search, self.precision = cmap_mode, search, precision
        self.res_layers = tuple(res_layers)
        if voxelizer not in ("auto", "hip", "reference"):
            raise IconAmdError("voxelizer must be 'auto', 'hip' or 'reference'")
        self.voxelizer = voxelizer       # pamir: which semantic voxeliser feeds netG.ve (see _pamir_volume)
        self.last_op = None              # for regressors given as a state_dict: None or "sigmoid" (modules carry their own last_op)
        self.tie_rule = None             # diagnostics: ("highest", ulps) - see Workspace.set_tie_rule / DESIGN.md section 2
        self.norm_mlp = None             # for regressors given as a state_dict: "group" / "instance" (modules say it themselves)
        self._calibrated = None          # (mlp key, precision) the effective precision was derived for
        self._work_tie = None
        self.netG = None
        self.work = None
        self._mesh = self._mesh_key = self._mesh_src = None
        self._feat = self._feat_key = self._feat_src = None
        self._mlp = self._mlp_key = self._mlp_src = None
        self._vol = self._vol_key = self._vol_src = None
        self._vol_cached = None
        self._smpl_feat_dict = None
        self._regressor = None

    # ---- binding -----------------------------------------------------------------------------
    @classmethod
    def attach(cls, netG, **kw) -> "IconQueryEngine":
        """Replace ``netG.query`` by the HIP path; everything else on netG is untouched."""
        # lib/net/HGPIFuNet.py:32 ``maskout = False`` is a MODULE constant read inside query() (:337-342: with it the sdf
        # channel of every point whose selected image features sum to exactly 0 is overwritten with -1).  Nothing upstream
        # sets it; a fork that does would get another smpl_feat from its own query() than from this one - refuse instead of
        # diverging silently
        import sys
        mod = sys.modules.get(type(netG).__module__)
        if mod is not None and getattr(mod, "maskout", False):
            raise IconAmdError(f"attach: {type(netG).__module__}.maskout is set (lib/net/HGPIFuNet.py:32,337-342 overwrites the sdf "
                               "channel of points without image features with it); the HIP query evaluates the shipped maskout = False path only")
        eng = cls(prior_type=netG.prior_type, sdf_clip=netG.sdf_clip,
                  smpl_feats=getattr(netG, "smpl_feats", ("sdf", "norm", "vis", "cmap")),
                  res_layers=getattr(netG.if_regressor, "res_layers", (2, 3, 4)), **kw)
        eng.netG = netG
        netG.query = eng.query
        netG.icon_amd_engine = eng
        return eng

    def set_mesh(self, smpl_verts, smpl_faces, smpl_cmap, smpl_vis) -> None:
        self._smpl_feat_dict = dict(smpl_verts=smpl_verts, smpl_faces=smpl_faces, smpl_cmap=smpl_cmap,
                                    smpl_vis=smpl_vis)

    def set_regressor(self, regressor) -> None:
        self._regressor = regressor

    def set_volume_features(self, vol_feat: torch.Tensor) -> None:
        """PaMIR: the VolumeEncoder output [1,Cv,D,H,W] (lib/net/HGPIFuNet.py:321-325), computed once
        per image on PyTorch-ROCm by the caller (hoisted out of query())."""
        self._vol = vol_feat

    # ---- handle caches ---------------------------------------------------------------------------
    split_features = True      # slab_features / slab_finish_gathered take ``work=1``: a second workspace, so that the two halves of a
                               # Z-slab can each run phase 1 and exchange their outlier signs independently (recon._forward_sharded)

    def _work(self, i: int = 0) -> Workspace:
        """workspace ``i`` of this engine (0: every ordinary call; 1: the second half-slab of the sharded protocol)"""
        if i == 0:
            if self.work is None:
                self.work = Workspace()
                self._work_tie = None
            w = self.work
        else:
            extra = self.__dict__.setdefault("_extra_works", {})
            if i not in extra:
                extra[i] = Workspace()
                extra[i]._tie = None
            w = extra[i]
        have = self._work_tie if i == 0 else w._tie
        if self.tie_rule != have:
            if self.tie_rule is None:
                w.set_tie_rule(0, 0)
            else:
                kind, ulps = self.tie_rule
                if kind != "highest":
                    raise IconAmdError("tie_rule must be None or ('highest', ulps)")
                w.set_tie_rule(1, int(ulps))
            if i == 0:
                self._work_tie = self.tie_rule
            else:
                w._tie = self.tie_rule
        return w

    def _mesh_handle(self) -> Optional[MeshHandle]:
        if self.prior_type != "icon":
            return None
        d = self.netG.smpl_feat_dict if self.netG is not None else self._smpl_feat_dict
        if d is None:
            raise IconAmdError("no SMPL tensors bound: call filter() on the network or set_mesh() first")
        ts = (d["smpl_verts"], d["smpl_faces"], d["smpl_cmap"], d["smpl_vis"])
        k = _key(*ts)
        if k != self._mesh_key:
            old = getattr(self, "_mesh", None)
            if old is not None and not old.checked and old.h:
                # apps bind new SMPL tensors per image and may issue ONE asynchronous call per mesh: a mesh nobody polled must not
                # be dropped unseen (the device build makes bad input harmless - vertex 0, coordinate 0 - and a plausible but wrong
                # volume would be all the caller ever gets).  Its work was enqueued a whole image ago: this wait is over.
                try:
                    old.status(wait=True)
                except IconAmdError as e:
                    self._mesh, self._mesh_key, self._mesh_src = None, None, None
                    raise IconAmdError(f"the PREVIOUS SMPL mesh bound to this engine was invalid (its results are wrong): {e}") from e
            self._mesh = MeshHandle(*ts, validate=False)
            self._mesh_key, self._mesh_src = k, ts      # strong refs: see _key
        elif not self._mesh.checked:
            self._mesh.status()                         # a host read of a pinned word: raises once the build has reported bad input
        return self._mesh

    def poll_mesh_status(self, wait: bool = False) -> None:
        """Raise if the device build of the bound mesh has reported bad input (face naming a missing vertex, non-finite
        coordinate).  A host read of a pinned word - callers invoke it wherever they have just synchronised (the counts of
        adaptive_eval, the None test of reconEngine.forward), so every mesh is checked by the end of ITS image."""
        m = getattr(self, "_mesh", None)
        if m is not None and not m.checked and m.h:
            m.status(wait)

    def poll_work_status(self) -> None:
        """Raise if a shared-walk search launched on one of this engine's workspaces gave up on a hand-over (icon_work_status:
        a host read of each workspace's error record).  Like poll_mesh_status, for callers that have just synchronised: the
        volume of THAT call is the suspect one - without this the record would only surface when the next call on the
        workspace refuses to start (query_kernels.hip: ensure_work), one image late."""
        works = [self.work] + list(self.__dict__.get("_extra_works", {}).values())
        for w in works:
            if w is not None and w.h:
                w.status()

    def mesh_z_range(self):
        """(z_min, z_max) of the bound SMPL vertices in world coordinates (one D2H read per mesh, cached):
        the cost model of the Z-slab cut (recon.plane_weights).  None for priors without a body mesh."""
        if self.prior_type != "icon":
            return None
        self._mesh_handle()
        if getattr(self, "_zr_key", None) != self._mesh_key:
            z = self._mesh_src[0].detach().reshape(-1, 3)[:, 2].float()
            self._zr = (float(z.min()), float(z.max()))
            self._zr_key = self._mesh_key
        return self._zr

    def _feat_handle(self, im_feat: torch.Tensor) -> FeatHandle:
        vol = self._pamir_volume() if self.prior_type == "pamir" else None
        k = _key(im_feat) + (_key(vol) if vol is not None else ())
        if k != self._feat_key:
            select = 2 if (self.prior_type == "icon" and "vis" in self.smpl_feats) else 1
            self._feat = FeatHandle(im_feat, select, vol, smpl_feats=self.smpl_feats)
            self._feat_key, self._feat_src = k, (im_feat, vol)
        return self._feat

    def _pamir_volume(self) -> torch.Tensor:
        """VolumeEncoder output [1,Cv,D,H,W] of the current image (lib/net/HGPIFuNet.py:314-325).  The
        reference voxelises and encodes on EVERY query() call; both only depend on the image, so they run once
        per set of voxel tensors: the semantic volume on the HIP voxeliser (semantic_voxelization - no
        voxelize_cuda wheel needed), the 3-D convolutions of ``netG.ve`` on PyTorch-ROCm."""
        if self._vol is not None:
            return self._vol
        netG = self.netG
        if netG is None or not hasattr(netG, "voxelization") or not hasattr(netG, "ve"):
            raise IconAmdError("pamir prior: call set_volume_features(vol_feat) (VolumeEncoder output)")
        d = netG.smpl_feat_dict
        k = _key(d["voxel_verts"], d["voxel_faces"])
        if k != self._vol_key:
            vv = d["voxel_verts"][:, :-int(d["pad_v_num"][0]), :]
            vf = d["voxel_faces"][:, :-int(d["pad_f_num"][0]), :]
            vox = netG.voxelization
            if self._use_reference_voxelizer():
                # the reference's own leaf (lib/net/voxelize.py:119-137 -> voxelize_cuda wheel), exactly as
                # HGPIFuNet.query calls it (lib/net/HGPIFuNet.py:320) - still once per image, not per query()
                with torch.no_grad():
                    vox.update_param(batch_size=vf.shape[0], smpl_tetra=vf[0].detach().cpu().numpy())   # HGPIFuNet.py:321-323
                    vol = vox(vv)                                                                        # :324, vol ~ [0,1]
            else:
                vol = semantic_voxelization(vv, vf, vox.smpl_vertex_code, res=int(getattr(vox, "volume_res", 128)),
                                            sigma=float(getattr(vox, "sigma", 0.05)))
            with torch.no_grad():
                self._vol_cached = netG.ve(vol, intermediate_output=False)[-1]
            self._vol_key, self._vol_src = k, (d["voxel_verts"], d["voxel_faces"])
        return self._vol_cached

    def _use_reference_voxelizer(self) -> bool:
        """voxelizer='reference': always netG.voxelization (needs the voxelize_cuda wheel); 'hip': always the HIP
        voxeliser; 'auto' (default): the reference's leaf when its wheel imports, otherwise the HIP voxeliser with ONE
        warning - its semantics are restated from the call site, not verified against voxelize_cuda (PARITY UNPINNED)."""
        if self.voxelizer == "hip":
            return False
        try:
            import voxelize_cuda
            have = hasattr(voxelize_cuda, "forward_semantic_voxelization")      # the entry lib/net/voxelize.py:57 calls
        except Exception:
            have = False
        if self.voxelizer == "reference":
            if not have:
                raise IconAmdError("voxelizer='reference' needs the voxelize_cuda wheel (requirements.txt:34), which does not import")
            return True
        if not have:
            global _WARNED_VOXELIZER
            if not _WARNED_VOXELIZER:
                import warnings
                warnings.warn("icon_amd: voxelize_cuda is not installed - the PaMIR semantic volume comes from the HIP voxeliser, whose "
                              "semantics are restated from lib/net/voxelize.py and NOT verified against the wheel (parity unpinned); "
                              "pass voxelizer='reference' to insist on the wheel, 'hip' to silence this")
                _WARNED_VOXELIZER = True
        return have

    def _bound_regressor(self, regressor=None):
        reg = regressor if regressor is not None else self._regressor
        if reg is None and self.netG is not None:
            reg = self.netG.if_regressor
        if reg is None:
            raise IconAmdError("no regressor bound: pass regressor= or call set_regressor()")
        return reg

    def _composed_reason(self, reg, im_feat) -> Optional[str]:
        """why this regressor / feature layout is outside what the fused kernels carry (icon_amd/composed.py evaluates it from
        the HIP geometry leaf + PyTorch-ROCm operators instead), or None"""
        from . import composed
        # cached per (regressor tensors, feature width): walking a weight-normed state_dict copies it to the host - not per query
        try:
            tens = list(reg.values()) if isinstance(reg, dict) else list(reg.parameters())
            ck = (tuple((id(t), getattr(t, "_version", 0)) for t in tens), int(im_feat.shape[-3]), self.last_op, self.res_layers,
                  self.smpl_feats, self.prior_type, id(getattr(reg, "last_op", None)))
        except Exception:
            ck = None
        if ck is not None and getattr(self, "_composed_key", None) == ck:
            return self._composed_val
        val = self._composed_reason_uncached(reg, im_feat)
        self._composed_key, self._composed_val, self._composed_src = ck, val, tens if ck is not None else None
        return val

    def _composed_reason_uncached(self, reg, im_feat) -> Optional[str]:
        from . import composed
        if isinstance(reg, dict):
            sd = effective_filters(reg)
            last_ok = self.last_op in (None, "sigmoid")
        else:
            sd = effective_filters(reg.state_dict())
            lo = getattr(reg, "last_op", None)
            last_ok = lo is None or isinstance(lo, nn.Sigmoid)
        n = 0
        while f"filters.{n}.weight" in sd:
            n += 1
        if n == 0:
            return None                                   # not an MLP state_dict: let the ordinary path say so
        shapes = [(int(sd[f"filters.{l}.weight"].shape[0]), int(sd[f"filters.{l}.weight"].shape[1])) for l in range(n)]
        C_ = int(im_feat.shape[-3])
        n_img = C_ // 2 if (self.prior_type == "icon" and "vis" in self.smpl_feats) else C_
        return composed.unsupported_reason(shapes, self.res_layers, last_ok, n_img, shapes[0][1])

    def _composed_query(self, reason, features, points, calibs, reg):
        from . import composed
        if reason not in _WARNED_COMPOSED:
            _WARNED_COMPOSED.add(reason)
            import warnings
            warnings.warn(f"icon_amd: {reason} - evaluated by the composed path (HIP geometry leaf + PyTorch-ROCm operators, "
                          "icon_amd/composed.py), about ten times the fused kernel's time")
        return composed.query_composed(self, features, points, calibs.to(points.device), reg)

    def _callnorm_spec(self, reg):
        """CallNormSpec of a Group / InstanceNorm regressor (icon_amd/callnorm.py), None for everything that folds"""
        from . import callnorm
        if isinstance(reg, dict):
            if self.norm_mlp not in ("group", "instance"):
                return None
            sd = effective_filters(reg)
        else:
            if getattr(reg, "norm", None) not in ("group", "instance") or len(getattr(reg, "norms", ())) == 0:
                return None
            sd = reg.state_dict()
        n = 0
        while f"filters.{n}.weight" in sd:
            n += 1
        return callnorm.spec_of(reg, self.norm_mlp, [int(sd[f"filters.{l}.weight"].shape[0]) for l in range(n - 1)])

    def _callnorm_eval(self, reg, spec, rows: torch.Tensor) -> torch.Tensor:
        """occupancy of the call whose MLP input rows are ``rows`` [N,16] under a Group / InstanceNorm regressor: the call's
        statistics (callnorm.call_statistics), folded like BatchNorm, the MFMA MLP on the rows, the in_cube mask"""
        from . import callnorm
        sd = effective_filters(regressor_state_dict(reg, self.norm_mlp))
        n = 0
        while f"filters.{n}.weight" in sd:
            n += 1
        dev = rows.device
        W = [torch.as_tensor(sd[f"filters.{l}.weight"]).detach().to(dev, torch.float32).reshape(sd[f"filters.{l}.weight"].shape[0], -1) for l in range(n)]
        b = [torch.as_tensor(sd[f"filters.{l}.bias"]).detach().to(dev, torch.float32) for l in range(n)]
        c0 = int(W[0].shape[1])
        is_res = [l in self.res_layers for l in range(n)]
        means, variances = callnorm.call_statistics(W, b, is_res, spec, rows, c0)
        last_op = regressor_last_op(reg) if not isinstance(reg, dict) else self.last_op
        handle = MlpHandle(callnorm.batchnorm_equivalent({k: torch.as_tensor(v).detach().cpu() for k, v in sd.items()}, spec, means, variances),
                           self.res_layers, last_op=last_op)
        occ = handle.forward(rows, "f32" if self.precision == "f32" else "f16x3")
        return occ * callnorm.in_cube_mask(rows)

    def _rows(self, im_feat, points=None, calib12=None, lattice=None) -> torch.Tensor:
        """the MLP input rows [N,16] of a call (icon_query_rows / icon_grid_rows): explicit ``points`` [N,3] with ``calib12``, or
        ``lattice = (res, z0, z1)``"""
        mesh, feat = self._mesh_handle(), self._feat_handle(im_feat)
        mh = mesh.h if mesh is not None else C.c_void_p(0)
        common = (C.c_int(_lib.PRIOR[self.prior_type]), C.c_float(np.float32(self.sdf_clip)), C.c_int(_lib.CMAP[self.cmap_mode]))
        if lattice is not None:
            res, z0, z1 = lattice
            rows = torch.empty(((z1 - z0) * res * res, 16), dtype=torch.float32, device=im_feat.device)
            check(_lib.lib().icon_grid_rows(mh, feat.h, *common, C.c_int(res), C.c_int(z0), C.c_int(z1), ptr(rows),
                                            C.c_int(_lib.SEARCH[self.search]), self._work().h, _stream()), "icon_grid_rows")
            return rows
        rows = torch.empty((points.shape[0], 16), dtype=torch.float32, device=points.device)
        on_dev = isinstance(calib12, torch.Tensor)
        check(_lib.lib().icon_query_rows(mh, feat.h, *common, C.c_void_p(0) if on_dev else ptr(calib12), ptr(calib12) if on_dev else C.c_void_p(0),
                                         ptr(points), C.c_int64(points.shape[0]), ptr(rows), C.c_int(_lib.SEARCH[self.search]),
                                         self._work().h, _stream()), "icon_query_rows")
        return rows

    def _mlp_handle(self, regressor=None) -> MlpHandle:
        reg = self._bound_regressor(regressor)
        if self._callnorm_spec(reg) is not None:
            raise IconAmdError("this regressor normalises over the points of the call (norm_mlp 'group' / 'instance'): it is evaluated by "
                               "query() and whole-lattice eval_slab() only, not through the split slab protocol (use shard=False)")
        sd = regressor_state_dict(reg)
        last_op = regressor_last_op(reg) if not isinstance(reg, dict) else self.last_op
        k = tuple((n, ) + (_key(t)[0] if isinstance(t, torch.Tensor) else (id(t),)) for n, t in sd.items()) + (last_op,)
        if k != self._mlp_key:
            self._mlp = MlpHandle(sd, self.res_layers, last_op=last_op)
            self._mlp_key, self._mlp_src = k, list(sd.values())
        self._resolve_precision()
        return self._mlp

    def _resolve_precision(self) -> None:
        """The precision the kernels run at for (current checkpoint, current ``self.precision``) - re-derived whenever either
        changes, so assigning ``eng.precision`` after the first query takes effect."""
        if self.precision not in _lib.PRECISION:
            raise IconAmdError(f"unknown precision {self.precision!r} (one of {sorted(_lib.PRECISION)})")
        self._effective_precision = self.precision
        self._calibrated = (self._mlp_key, self.precision)

    def _precision(self) -> int:
        if self._mlp is not None:
            self._resolve_precision()
        return _lib.PRECISION[getattr(self, "_effective_precision", self.precision)]

    # ---- HGPIFuNet.query ---------------------------------------------------------------------------
    @_guarded
    def query(self, features, points, calibs, transforms=None, regressor=None):
        """features: list of [1,C,H,W]; points [1,3,N]; calibs [1,4,4] (or [1,3,4]) -> list of [1,1,N]"""
        if points.dim() != 3 or points.shape[0] != 1 or points.shape[1] != 3:
            raise IconAmdError("points must be [1,3,N] (batch size 1)")
        if not points.is_cuda:
            raise IconAmdError("points are on the CPU; icon_amd has no CPU path")
        n = int(points.shape[2])
        if n == 0:
            return [torch.empty((1, 1, 0), dtype=torch.float32, device=points.device) for _ in features]
        if transforms is not None:
            # lib/net/geometry.py:57-60 indexes `transforms[:2, :2]` / `transforms[:2, 2:3]` and feeds the slices to
            # torch.baddbmm: for the documented [B,2,3] layout the shift slice is empty and for a [2,3] matrix the
            # operands are 2-D - the reference raises for every shape, and no caller in the tree passes the
            # argument (tests/test_oracle_vs_reference.py pins that).  There is no behaviour to reproduce.
            raise IconAmdError("query(transforms=...) is not supported: the reference's own orthogonal() raises for any "
                               "`transforms` (lib/net/geometry.py:57-60) and none of its callers passes one")
        if calibs.is_cuda:
            # stays on the device: the kernels read the 12 floats themselves (no D2H copy / stream sync per query)
            calib12 = calibs[0, :3, :4].detach().to(torch.float32).contiguous()
            pts = points[0].t().to(torch.float32).contiguous()
        else:
            calib12 = np.ascontiguousarray(calibs[0, :3, :4].detach().to(torch.float32).numpy())
            pts = points[0].t().to(torch.float32).contiguous()
        reg = self._bound_regressor(regressor)
        reason = self._composed_reason(reg, features[-1]) if len(features) else None
        if reason is not None:      # outside what the kernels carry: the reference's operator sequence around the HIP geometry leaf
            return self._composed_query(reason, features, points, calibs, reg)
        spec = self._callnorm_spec(reg)
        if spec is not None:        # Group / InstanceNorm: the statistics of THIS call's points (icon_amd/callnorm.py)
            return [self._callnorm_eval(reg, spec, self._rows(im_feat, points=pts, calib12=calib12)).view(1, 1, n) for im_feat in features]
        mesh = self._mesh_handle()
        mlp = self._mlp_handle(regressor)
        preds = []
        for im_feat in features:
            feat = self._feat_handle(im_feat)
            occ = torch.empty(n, dtype=torch.float32, device=points.device)
            fn = _lib.lib().icon_query_points_dcalib if isinstance(calib12, torch.Tensor) else _lib.lib().icon_query_points
            check(fn(
                mesh.h if mesh is not None else C.c_void_p(0), feat.h, mlp.h, C.c_int(_lib.PRIOR[self.prior_type]),
                C.c_float(np.float32(self.sdf_clip)), C.c_int(_lib.CMAP[self.cmap_mode]),
                ptr(calib12) if calib12 is not None else C.c_void_p(0), ptr(pts), C.c_int64(n), ptr(occ),
                C.c_int(_lib.SEARCH[self.search]), C.c_int(self._precision()), self._work().h, _stream()),
                "icon_query_points")
            preds.append(occ.view(1, 1, n))
        return preds

    # ---- dense lattice (one rank's share of reconEngine) ------------------------------------------------
    @_guarded
    def eval_slab(self, im_feat, res: int, z0: int, z1: int, regressor=None, out=None) -> torch.Tensor:
        reg = self._bound_regressor(regressor)
        reason = self._composed_reason(reg, im_feat)
        if reason is not None:      # the lattice materialised exactly as batch_eval does, one composed query over the planes [z0,z1)
            from .recon import lattice_coords
            dev_ = im_feat.device
            b_min = torch.tensor([[-1.0, 1.0, -1.0]], device=dev_)
            b_max = torch.tensor([[1.0, -1.0, 1.0]], device=dev_)
            pts_l = lattice_coords(res, b_min, b_max, True, dev_)[:, z0 * res * res: z1 * res * res]
            occ = self._composed_query(reason, [im_feat], pts_l.permute(0, 2, 1).contiguous(), torch.eye(4, device=dev_)[None], reg)[0]
            occ = occ.view(z1 - z0, res, res)
            if out is not None:
                out.copy_(occ)
                return out
            return occ
        spec = self._callnorm_spec(reg)
        if spec is not None:        # Group / InstanceNorm: the planes [z0,z1) are ONE call, their points the statistics' population
            occ = self._callnorm_eval(reg, spec, self._rows(im_feat, lattice=(res, z0, z1))).view(z1 - z0, res, res)
            if out is not None:
                out.copy_(occ)
                return out
            return occ
        mesh, mlp, feat = self._mesh_handle(), self._mlp_handle(regressor), self._feat_handle(im_feat)
        if out is None:
            out = torch.empty((z1 - z0, res, res), dtype=torch.float32, device=im_feat.device)
        check(_lib.lib().icon_grid_eval_slab(
            mesh.h if mesh is not None else C.c_void_p(0), feat.h, mlp.h, C.c_int(_lib.PRIOR[self.prior_type]),
            C.c_float(np.float32(self.sdf_clip)), C.c_int(_lib.CMAP[self.cmap_mode]), C.c_int(res), C.c_int(z0),
            C.c_int(z1), ptr(out), C.c_int(_lib.SEARCH[self.search]), C.c_int(self._precision()),
            self._work().h, _stream()), "icon_grid_eval_slab")
        return out

    def native_schedule_reason(self, im_feat, regressor=None, _handle_out: Optional[list] = None) -> Optional[str]:
        """why the reference's coarse-to-fine schedule cannot run as ONE native call (icon_adaptive_eval) for the bound
        regressor / settings - the host-driven schedule of recon.AdaptiveReconEngine takes over then - or None"""
        reg = self._bound_regressor(regressor)
        if self._composed_reason(reg, im_feat) is not None or self._callnorm_spec(reg) is not None:
            return "regressor outside what the fused kernels carry"
        if self.search != "bvh":
            return "search != 'bvh'"
        if self.tie_rule is not None:
            return "diagnostics tie rule set"
        mlp = self._mlp_handle(regressor)
        if getattr(self, "_effective_precision", self.precision) != "f16x3":
            return f"precision {getattr(self, '_effective_precision', self.precision)!r}"
        if _handle_out is not None:              # the caller runs adaptive_eval(_mlp=...) right behind this check: the key over the
            _handle_out.append(mlp)              # checkpoint's 21 tensors is computed once per call, not twice
        return None

    @_guarded
    def adaptive_eval(self, im_feat, resolutions: Sequence[int], balance: float = 0.5, regressor=None, counts: bool = True, _mlp=None):
        """Seg3dLossless._forward_faster (lib/common/seg3d_lossless.py:152-265) as one native call (icon_adaptive_eval): ->
        (volume [R,R,R] of the last resolution, points queried per level, bool: some voxel of the coarsest level > 0.5).
        ``counts=False``: nothing is read back (the last two are None; Workspace counters hold them)."""
        mesh, mlp, feat = self._mesh_handle(), (_mlp if _mlp is not None else self._mlp_handle(regressor)), self._feat_handle(im_feat)
        res = [int(r) for r in resolutions]
        n = len(res)
        out = torch.empty((res[-1],) * 3, dtype=torch.float32, device=im_feat.device)
        arr = (C.c_int * n)(*res)
        hc = (C.c_int64 * (n + 1))()
        check(_lib.lib().icon_adaptive_eval(
            mesh.h if mesh is not None else C.c_void_p(0), feat.h, mlp.h, C.c_int(_lib.PRIOR[self.prior_type]),
            C.c_float(np.float32(self.sdf_clip)), C.c_int(_lib.CMAP[self.cmap_mode]), arr, C.c_int(n), C.c_float(np.float32(balance)),
            ptr(out), hc if counts else None, C.c_int(_lib.SEARCH[self.search]), C.c_int(self._precision()), self._work().h, _stream()),
            "icon_adaptive_eval")
        if not counts:
            return out, None, None
        self.poll_mesh_status()                         # (the call has synchronised for the counts: the mesh build has reported)
        return out, [int(hc[k]) for k in range(n)], bool(hc[n])

    @_guarded
    def slab_features(self, im_feat, res: int, z0: int, z1: int, signs=None, count=None, msg=None, work: int = 0):
        """Phase 1 of the split protocol -> (signs int8 [cap] device, count int64 [1] device).  With ``msg`` (a
        contiguous int8 / uint8 device buffer of >= 8 + ceil(points / 4) bytes, e.g. this rank's slot of an all_gather
        input) the slab's exchange message [int64 K][2-bit packed signs] is written there instead and ``msg`` is returned."""
        mesh, feat = self._mesh_handle(), self._feat_handle(im_feat)
        n = (z1 - z0) * res * res
        if msg is not None:
            if not msg.is_contiguous() or msg.element_size() != 1 or msg.numel() < 8 + (n + 3) // 4:
                raise IconAmdError("slab_features: msg must be a contiguous byte buffer of >= 8 + ceil(points / 4) bytes")
            check(_lib.lib().icon_grid_slab_features_msg(
                mesh.h if mesh is not None else C.c_void_p(0), feat.h, C.c_int(_lib.PRIOR[self.prior_type]),
                C.c_float(np.float32(self.sdf_clip)), C.c_int(_lib.CMAP[self.cmap_mode]), C.c_int(res), C.c_int(z0),
                C.c_int(z1), ptr(msg), C.c_int64(msg.numel()), C.c_int(_lib.SEARCH[self.search]), self._work(work).h, _stream()),
                "icon_grid_slab_features_msg")
            return msg
        if signs is None:
            signs = torch.empty(n, dtype=torch.int8, device=im_feat.device)
        if count is None:
            count = torch.zeros(1, dtype=torch.int64, device=im_feat.device)
        if signs.numel() < n or signs.dtype != torch.int8 or count.dtype != torch.int64 or not signs.is_contiguous():
            raise IconAmdError("slab_features: signs must be a contiguous int8 buffer of >= slab points, count int64")
        check(_lib.lib().icon_grid_slab_features(
            mesh.h if mesh is not None else C.c_void_p(0), feat.h, C.c_int(_lib.PRIOR[self.prior_type]),
            C.c_float(np.float32(self.sdf_clip)), C.c_int(_lib.CMAP[self.cmap_mode]), C.c_int(res), C.c_int(z0),
            C.c_int(z1), ptr(signs), ptr(count), C.c_int(_lib.SEARCH[self.search]), self._work(work).h, _stream()),
            "icon_grid_slab_features")
        return signs, count

    @_guarded
    def slab_finish(self, res: int, z0: int, z1: int, signs_global, k_total: int, rank_offset: int,
                    regressor=None, out=None, device=None) -> torch.Tensor:
        mlp = self._mlp_handle(regressor)
        if out is None:
            out = torch.empty((z1 - z0, res, res), dtype=torch.float32,
                              device=device if device is not None else signs_global.device)
        check(_lib.lib().icon_grid_slab_finish(
            mlp.h, C.c_int(res), C.c_int(z0), C.c_int(z1), ptr(signs_global) if signs_global is not None else C.c_void_p(0),
            C.c_int64(k_total), C.c_int64(rank_offset), ptr(out), C.c_int(self._precision()),
            self._work().h, _stream()), "icon_grid_slab_finish")
        return out


    @_guarded
    def slab_finish_gathered(self, res: int, z0: int, z1: int, gathered: Optional[torch.Tensor], stride: int, world: int, rank: int,
                             regressor=None, out=None, za: Optional[int] = None, zb: Optional[int] = None, device=None, work: int = 0) -> torch.Tensor:
        """Phase 2 on the all_gather output itself: ``gathered`` bytes [world * stride], message r =
        [int64 count_r][2-bit packed signs_r] (slab_features(msg=...)); nothing is read back to the host.
        ``gathered=None``: no exchange (cmap_mode local / one rank).  Evaluates the planes [za, zb) of the slab
        (default: all of it) into ``out`` [(z1-z0), res, res], the SLAB's buffer; may be called piece by piece.  ``work``: the
        workspace phase 1 of THIS slab ran on (slab_features(work=...))."""
        mlp = self._mlp_handle(regressor)
        if out is None:
            out = torch.empty((z1 - z0, res, res), dtype=torch.float32, device=gathered.device if gathered is not None else device)
        za = z0 if za is None else za
        zb = z1 if zb is None else zb
        check(_lib.lib().icon_grid_slab_finish_gathered(
            mlp.h, C.c_int(res), C.c_int(z0), C.c_int(z1), C.c_int(za), C.c_int(zb), ptr(gathered), C.c_int64(stride),
            C.c_int(world), C.c_int(rank), ptr(out), C.c_int(self._precision()), self._work(work).h, _stream()),
            "icon_grid_slab_finish_gathered")
        return out


def query_func(opt, netG, features, points, proj_matrix=None):
    """lib/common/train_util.py:324-348, verbatim contract: points [1,N,3] -> [1,1,N].
    ``netG`` is either a reference HGPIFuNet with an attached engine or an IconQueryEngine."""
    assert len(points) == 1
    if getattr(opt, "num_views", 1) != 1:
        raise IconAmdError("num_views must be 1 (lib/common/config.py:34)")
    samples = points.permute(0, 2, 1)  # [1,3,N]
    if proj_matrix is not None:
        rot, trans = proj_matrix[:, :3, :3], proj_matrix[:, :3, 3:4]
        samples = torch.baddbmm(trans, rot, samples)
    calib = torch.eye(4, dtype=torch.float32, device=samples.device)[None]
    regressor = getattr(netG, "if_regressor", None)
    preds = netG.query(features=features, points=samples, calibs=calib, regressor=regressor)
    if type(preds) is list:
        preds = preds[0]
    return preds


def semantic_voxelization(voxel_verts: torch.Tensor, voxel_tets: torch.Tensor, vertex_code, res: int = 128,
                          sigma: float = 0.05) -> torch.Tensor:
    """Drop-in for ``Voxelization.forward`` (lib/net/voxelize.py:119-137 -> voxelize_cuda, an external CUDA
    wheel): ``voxel_verts [1,V,3]`` (tetrahedralised SMPL in the [-0.5,0.5] cube, surface vertices first;
    lib/dataset/TestDataset.py:150-192 with the padding stripped as lib/net/HGPIFuNet.py:316-319 does),
    ``voxel_tets [1,T,4]`` vertex indices, ``vertex_code [Vs,3]`` the semantic code of the surface vertices
    (``Voxelization.smpl_vertex_code``) -> ``[1,3,res,res,res]`` (b,c,d,h,w), what ``vol.permute(0,4,1,2,3)`` returns.
    PARITY UNPINNED: see include/icon_amd.h (icon_semantic_voxelize)."""
    if not voxel_verts.is_cuda:
        raise IconAmdError("semantic_voxelization needs device tensors (there is no CPU path)")
    dev = voxel_verts.device
    v = voxel_verts.detach().to(torch.float32).reshape(-1, 3).contiguous()
    t = voxel_tets.detach().to(dev, torch.int64).reshape(-1, 4).contiguous()
    code = torch.as_tensor(np.asarray(vertex_code, dtype=np.float32) if not isinstance(vertex_code, torch.Tensor) else vertex_code)
    code = code.detach().to(dev, torch.float32).reshape(-1, 3).contiguous()
    if code.shape[0] > v.shape[0]:
        raise IconAmdError("semantic_voxelization: more vertex codes than vertices")
    out = torch.empty((res, res, res, 3), dtype=torch.float32, device=dev)
    check(_lib.lib().icon_semantic_voxelize(ptr(v), C.c_int64(v.shape[0]), C.c_int64(code.shape[0]), ptr(code), ptr(t),
                                            C.c_int64(t.shape[0]), C.c_int(res), C.c_float(sigma), ptr(out), _stream()),
          "icon_semantic_voxelize")
    return out.permute(3, 0, 1, 2).unsqueeze(0)


def get_visibility(xy: torch.Tensor, z: torch.Tensor, faces: torch.Tensor, image_size: int = 2 ** 12) -> torch.Tensor:
    """Drop-in for ``lib.dataset.mesh_util.get_visibility`` (mesh_util.py:280-316): ``xy [N,2]``,
    ``z [N,1]``, ``faces [F,3]`` -> ``vis_mask [N,1]`` float32 in {0,1} (1 = the vertex belongs to a
    face that owns a pixel of the 4096^2 orthographic rasterisation; ``faces[-1]`` is always marked,
    as in the reference).  Inputs may live on the host (the reference's call sites pass CPU tensors,
    TestDataset.py:134-137) - they are moved to the current HIP device; the result comes back on
    the device of ``z``.  No pytorch3d involved."""
    if not torch.cuda.is_available():
        raise IconAmdError("get_visibility needs the HIP device (there is no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    out_dev = z.device
    xy_d = xy.detach().to(dev, torch.float32).reshape(-1, 2).contiguous()
    z_d = z.detach().to(dev, torch.float32).reshape(-1).contiguous()
    f_d = faces.detach().to(dev, torch.int64).reshape(-1, 3).contiguous()
    if xy_d.shape[0] != z_d.shape[0]:
        raise IconAmdError("get_visibility: xy and z disagree on the vertex count")
    if f_d.numel() and (int(f_d.min()) < 0 or int(f_d.max()) >= z_d.shape[0]):
        raise IconAmdError("get_visibility: face index out of range")
    vis = torch.empty(z_d.shape[0], device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().icon_visibility(C.c_void_p(xy_d.data_ptr()), C.c_void_p(z_d.data_ptr()), C.c_int64(z_d.shape[0]),
                                          C.c_void_p(f_d.data_ptr()), C.c_int64(f_d.shape[0]), C.c_int(int(image_size)),
                                          C.c_void_p(vis.data_ptr()), _stream()), "icon_visibility")
    return vis[:, None].to(out_dev)


def compute_vis_cmap(smpl_verts, smpl_faces, cmap_table, smplx_ind=None, device=None) -> dict:
    """Drop-in for ``TestDataset.compute_vis_cmap`` (lib/dataset/TestDataset.py:134-148): the per-image SMPL tensors that
    ``HGPIFuNet.filter`` copies into ``smpl_feat_dict`` (lib/net/HGPIFuNet.py:236-240).  ``smpl_verts [V,3]`` in the [-1,1]
    cube, ``smpl_faces [F,3]``; ``cmap_table`` is what ``SMPLX.get_smpl_mat`` loads (data/smpl_related/smpl_data/
    smplx_cmap.npy - an asset, passed in), ``smplx_ind`` the SMPL->SMPL-X vertex map for ``smpl_type == 'smpl'``
    (``SMPLX.smpl2smplx``; default: identity, the 'smplx' branch).  Visibility comes from the HIP z-buffer
    (get_visibility) - no pytorch3d.  Returns the reference's dict: smpl_vis [1,V,1], smpl_cmap [1,V,3] on ``device``,
    smpl_verts [1,V,3] as passed."""
    v = torch.as_tensor(smpl_verts)
    f = torch.as_tensor(smpl_faces).long()
    if device is None:
        device = v.device if v.is_cuda else torch.device("cuda", torch.cuda.current_device())
    (xy, z) = v.split([2, 1], dim=1)
    smpl_vis = get_visibility(xy, -z, f)
    ind = torch.arange(smpl_vis.shape[0]) if smplx_ind is None else torch.as_tensor(np.asarray(smplx_ind)).long().reshape(-1)
    table = torch.as_tensor(np.asarray(cmap_table) if not isinstance(cmap_table, torch.Tensor) else cmap_table).float()
    smpl_cmap = table[ind.to(table.device), :]
    return {"smpl_vis": smpl_vis.unsqueeze(0).to(device), "smpl_cmap": smpl_cmap.unsqueeze(0).to(device),
            "smpl_verts": v.unsqueeze(0)}
