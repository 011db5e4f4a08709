"""ctypes front-end of oracle/icon_oracle.c - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under icon_amd/ does (tests/test_no_oracle_in_product.py enforces it).

Each function takes/returns numpy arrays with the reference's tensor shapes minus the batch
dimension (batch size is 1 on this path, lib/common/seg3d_lossless.py:73).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("icon_oracle.c", "icon_accel.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = C.CDLL(_SO)
        except OSError:
            build(force=True)
            _lib = C.CDLL(_SO)
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_point_tri_dist2.restype = C.c_float
        _lib.orc_ray_hit.restype = C.c_int
        _lib.orc_get_accel.restype = C.c_int
        _lib.orc_accel_build.restype = C.c_void_p
        _lib.orc_icon_c0.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(C.c_int(n))


def vertex_normals(verts, faces):
    verts, faces = _f32(verts).reshape(-1, 3), _i64(faces).reshape(-1, 3)
    out = np.empty_like(verts)
    lib().orc_vertex_normals(_p(verts), C.c_int64(len(verts)), _p(faces), C.c_int64(len(faces)), _p(out))
    return out


def point_tri_dist2(p, a, b, c) -> float:
    p, a, b, c = (_f32(t).reshape(3) for t in (p, a, b, c))
    return float(lib().orc_point_tri_dist2(_p(p), _p(a), _p(b), _p(c)))


def nearest_brute(verts, faces, pts):
    verts, faces, pts = _f32(verts).reshape(-1, 3), _i64(faces).reshape(-1, 3), _f32(pts).reshape(-1, 3)
    d2 = np.empty(len(pts), np.float32)
    idx = np.empty(len(pts), np.int64)
    lib().orc_nearest_brute(_p(verts), _p(faces), C.c_int64(len(faces)), _p(pts), C.c_int64(len(pts)),
                            _p(d2), _p(idx))
    return d2, idx


def set_accel(on: bool) -> None:
    """cal_sdf / query_icon through the BVH + ray bins of icon_accel.c (default) or the linear scans"""
    lib().orc_set_accel(C.c_int(int(on)))


class Accel:
    """BVH + (y,z) ray bins over one mesh (icon_accel.c): the two O(N*F) leaves, bit-identical to the scans."""

    def __init__(self, verts, faces):
        self.verts, self.faces = _f32(verts).reshape(-1, 3), _i64(faces).reshape(-1, 3)
        self.h = C.c_void_p(lib().orc_accel_build(_p(self.verts), C.c_int64(len(self.verts)), _p(self.faces),
                                                  C.c_int64(len(self.faces))))

    def __del__(self):
        try:
            if self.h:
                lib().orc_accel_free(self.h)
                self.h = None
        except Exception:
            pass

    def nearest(self, pts):
        pts = _f32(pts).reshape(-1, 3)
        d2, idx = np.empty(len(pts), np.float32), np.empty(len(pts), np.int64)
        lib().orc_accel_nearest(self.h, _p(pts), C.c_int64(len(pts)), _p(d2), _p(idx))
        return d2, idx

    def check_sign(self, pts):
        pts = _f32(pts).reshape(-1, 3)
        out = np.empty(len(pts), np.uint8)
        lib().orc_accel_check_sign(self.h, _p(pts), C.c_int64(len(pts)), _p(out))
        return out.astype(bool)

    def nearest_ties(self, pts):
        """-> (d2, face, runner-up face (-1: none), ulps between their squared distances clipped to 255)"""
        pts = _f32(pts).reshape(-1, 3)
        n = len(pts)
        d2, idx, idx2, ulps = np.empty(n, np.float32), np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.uint8)
        lib().orc_accel_nearest_ties(self.h, _p(pts), C.c_int64(n), _p(d2), _p(idx), _p(idx2), _p(ulps))
        return d2, idx, idx2, ulps


def set_smpl_feats(has_cmap: bool = True, has_norm: bool = True, has_vis: bool = True) -> None:
    """cfg.net.smpl_feats for query_icon (lib/net/HGPIFuNet.py:301-309, :334-346): which of cmap / norm follow the sdf in the MLP
    input, and whether smpl_vis selects the feature half (without 'vis' every feature channel is an input)"""
    lib().orc_set_smpl_feats(C.c_int(int(has_cmap)), C.c_int(int(has_norm)), C.c_int(int(has_vis)))


def set_tie_rule(rule: int = 0, ulps: int = 0) -> None:
    """diagnostics: 1 = among the faces within ``ulps`` float32 ulps of the minimum d^2 the HIGHEST index wins
    (cal_sdf / query_icon through the accelerated leaves); 0 = the definition (lowest index on exact ties)"""
    lib().orc_set_tie_rule(C.c_int(rule), C.c_int(ulps))


def check_sign(verts, faces, pts):
    verts, faces, pts = _f32(verts).reshape(-1, 3), _i64(faces).reshape(-1, 3), _f32(pts).reshape(-1, 3)
    out = np.empty(len(pts), np.uint8)
    lib().orc_check_sign(_p(verts), _p(faces), C.c_int64(len(faces)), _p(pts), C.c_int64(len(pts)), _p(out))
    return out.astype(bool)


def cal_sdf(verts, faces, cmap, vis, pts):
    """-> dict(sdf [N], norm [N,3], cmap [N,3], vis [N], idx [N] int64, inside [N] bool)"""
    verts, faces = _f32(verts).reshape(-1, 3), _i64(faces).reshape(-1, 3)
    cmap, vis, pts = _f32(cmap).reshape(-1, 3), _f32(vis).reshape(-1), _f32(pts).reshape(-1, 3)
    n = len(pts)
    sdf, nrm, cm, vs = (np.empty(n, np.float32), np.empty((n, 3), np.float32),
                        np.empty((n, 3), np.float32), np.empty(n, np.float32))
    idx, ins = np.empty(n, np.int64), np.empty(n, np.uint8)
    lib().orc_cal_sdf(_p(verts), C.c_int64(len(verts)), _p(faces), C.c_int64(len(faces)), _p(cmap), _p(vis),
                      _p(pts), C.c_int64(n), _p(sdf), _p(nrm), _p(cm), _p(vs), _p(idx), _p(ins))
    return dict(sdf=sdf, norm=nrm, cmap=cm, vis=vs, idx=idx, inside=ins.astype(bool))


class _OrcMlp(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("cin", C.c_void_p), ("cout", C.c_void_p), ("is_res", C.c_void_p),
                ("W", C.c_void_p), ("b", C.c_void_p), ("bn_g", C.c_void_p), ("bn_b", C.c_void_p),
                ("bn_m", C.c_void_p), ("bn_v", C.c_void_p), ("last_op", C.c_int)]


def effective_filters(state_dict: dict) -> dict:
    """nn.utils.weight_norm (norm_mlp = 'weight', lib/net/MLP.py:42-45) stores filters.l.weight_g [Cout,1,1] and weight_v
    [Cout,Cin,1]; the weight the layer applies is v * (g / ||v||), the norm over everything but the output dimension"""
    if not any(k.endswith("weight_g") for k in state_dict):
        return state_dict
    out = {k: v for k, v in state_dict.items() if not (k.endswith("weight_g") or k.endswith("weight_v"))}
    for k in state_dict:
        if k.endswith("weight_g"):
            g = np.asarray(state_dict[k], np.float32)
            v = np.asarray(state_dict[k[:-1] + "v"], np.float32)
            nrm = np.sqrt((v.reshape(len(v), -1) ** 2).sum(1, dtype=np.float32)).reshape(g.shape)
            out[k[:-2]] = v * (g / nrm)
    return out


class Mlp:
    """Holds a reference-layout state_dict (numpy) as the orc_mlp struct."""

    def __init__(self, state_dict: dict, res_layers=(2, 3, 4), last_op=None):
        """last_op: None (cfg.test_mode) or "sigmoid" (lib/net/HGPIFuNet.py:133)"""
        state_dict = effective_filters(state_dict)
        n = 0
        while f"filters.{n}.weight" in state_dict:
            n += 1
        self.n = n
        self._keep = []
        W = [_f32(np.asarray(state_dict[f"filters.{l}.weight"]).reshape(
            np.asarray(state_dict[f"filters.{l}.weight"]).shape[0], -1)) for l in range(n)]
        b = [_f32(state_dict[f"filters.{l}.bias"]) for l in range(n)]
        self.c0 = W[0].shape[1]
        self.c_last = W[-1].shape[0]
        cin = np.array([w.shape[1] for w in W], np.int32)
        cout = np.array([w.shape[0] for w in W], np.int32)
        is_res = np.array([1 if l in res_layers else 0 for l in range(n)], np.int32)
        for l in range(n):
            expect = (cout[l - 1] if l else self.c0) + (self.c0 if is_res[l] else 0)
            assert cin[l] == expect, f"layer {l}: Cin {cin[l]} != {expect}"

        def ptr_array(arrs):
            arr = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
            self._keep.append((arrs, arr))
            return C.cast(arr, C.c_void_p)

        if "norms.0.running_mean" in state_dict:
            bn = {k: [_f32(state_dict[f"norms.{l}.{k}"]) for l in range(n - 1)]
                  for k in ("weight", "bias", "running_mean", "running_var")}
            bn_ptrs = [ptr_array(bn[k]) for k in ("weight", "bias", "running_mean", "running_var")]
        else:                                       # norm_mlp = 'weight' (or none): no norm layers, MLP.py:42-45,64-65
            bn, bn_ptrs = None, [C.c_void_p(0)] * 4
        self._keep += [cin, cout, is_res, W, b, bn]
        self.struct = _OrcMlp(n, cin.ctypes.data, cout.ctypes.data, is_res.ctypes.data,
                              ptr_array(W), ptr_array(b), *bn_ptrs, 1 if last_op == "sigmoid" else 0)

    def forward(self, x, f64: bool = False):
        """x [N, c0] point-major -> [N, c_last]"""
        x = _f32(x)
        assert x.ndim == 2 and x.shape[1] == self.c0
        out = np.empty((len(x), self.c_last), np.float32)
        lib().orc_mlp_forward(C.byref(self.struct), _p(x), C.c_int64(len(x)), C.c_int(self.c0), _p(out),
                              C.c_int(int(f64)))
        return out


class CallNormMlp:
    """MLP.forward (lib/net/MLP.py:49-72) with ``norm='group'`` (nn.GroupNorm(32, C), :35-36) or ``'instance'``
    (nn.InstanceNorm1d(C), :39-41): after every hidden Conv1d the output [1,C,N] is normalised with the mean / biased variance
    over (the channels of a group) x (all N points of the call) - InstanceNorm: one channel per group, no affine parameters
    unless the state_dict carries them - then LeakyReLU(0.01).  The whole call at once, numpy; statistics in float64."""

    def __init__(self, state_dict: dict, kind: str, res_layers=(2, 3, 4), last_op=None, groups: int = 32, eps: float = 1e-5):
        assert kind in ("group", "instance")
        sd = effective_filters(state_dict)
        n = 0
        while f"filters.{n}.weight" in sd:
            n += 1
        self.n, self.kind, self.groups, self.eps, self.res_layers, self.last_op = n, kind, groups, eps, tuple(res_layers), last_op
        self.W = [_f32(np.asarray(sd[f"filters.{l}.weight"]).reshape(np.asarray(sd[f"filters.{l}.weight"]).shape[0], -1)) for l in range(n)]
        self.b = [_f32(sd[f"filters.{l}.bias"]) for l in range(n)]
        self.c0 = self.W[0].shape[1]
        affine = "norms.0.weight" in sd
        self.gamma = [_f32(sd[f"norms.{l}.weight"]) if affine else None for l in range(n - 1)]
        self.beta = [_f32(sd[f"norms.{l}.bias"]) if affine else None for l in range(n - 1)]

    def plain(self) -> "Mlp":
        """the same filters without any norm (only good for asking the C oracle for the MLP input rows)"""
        sd = {f"filters.{l}.weight": self.W[l][:, :, None] for l in range(self.n)}
        sd.update({f"filters.{l}.bias": self.b[l] for l in range(self.n)})
        return Mlp(sd, self.res_layers)

    def forward(self, x) -> np.ndarray:
        """x [N, c0] (the whole call) -> [N]"""
        x = _f32(x)
        assert x.ndim == 2 and x.shape[1] == self.c0 and len(x) > 0
        N = len(x)
        h = x
        for l in range(self.n):
            inp = np.concatenate([h, x], 1) if l in self.res_layers else h
            y = inp @ self.W[l].T + self.b[l]
            if l == self.n - 1:
                h = y
                break
            C_ = y.shape[1]
            G = self.groups if self.kind == "group" else C_
            yg = y.reshape(N, G, C_ // G).astype(np.float64)
            mu = yg.mean(axis=(0, 2))
            var = yg.var(axis=(0, 2))                          # biased, as both torch norms use
            y = ((yg - mu[None, :, None]) / np.sqrt(var + self.eps)[None, :, None]).reshape(N, C_).astype(np.float32)
            if self.gamma[l] is not None:
                y = y * self.gamma[l] + self.beta[l]
            h = np.where(y < 0, np.float32(0.01) * y, y).astype(np.float32)
        out = h[:, 0]
        if self.last_op == "sigmoid":
            out = (1.0 / (1.0 + np.exp(-out.astype(np.float64)))).astype(np.float32)
        return out


_IDENT = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)


def _calib12(calib):
    if calib is None:
        return _IDENT.copy()
    c = _f32(calib).reshape(-1, 4)[:3]
    return np.ascontiguousarray(c)


def query_icon(verts, faces, cmap, vis, feat, mlp: Mlp, pts, sdf_clip=0.05, calib=None, f64=False,
               cmap_local=False):
    """HGPIFuNet.query, icon branch -> (occ [N], X [N, C/2+7]).  cmap_local=False reproduces the
    reference's tiled outlier-cmap assignment (HGPIFuNet.py:303-305), True the per-point rule."""
    verts, faces = _f32(verts).reshape(-1, 3), _i64(faces).reshape(-1, 3)
    cmap, vis, pts = _f32(cmap).reshape(-1, 3), _f32(vis).reshape(-1), _f32(pts).reshape(-1, 3)
    feat = _f32(feat)
    feat = feat.reshape(feat.shape[-3:])
    Cc, H, W = feat.shape
    n = len(pts)
    occ = np.empty(n, np.float32)
    X = np.empty((n, int(lib().orc_icon_c0(C.c_int(Cc)))), np.float32)
    cal = _calib12(calib)
    lib().orc_query_icon(_p(verts), C.c_int64(len(verts)), _p(faces), C.c_int64(len(faces)), _p(cmap), _p(vis),
                         _p(feat), C.c_int(Cc), C.c_int(H), C.c_int(W), C.byref(mlp.struct),
                         C.c_float(np.float32(sdf_clip)), _p(cal), _p(pts), C.c_int64(n), _p(occ), _p(X),
                         C.c_int(int(f64)), C.c_int(int(cmap_local)))
    return occ, X


def query_icon_callnorm(verts, faces, cmap, vis, feat, mlp: "CallNormMlp", pts, sdf_clip=0.05, calib=None, cmap_local=False):
    """HGPIFuNet.query (icon branch) with a Group / InstanceNorm regressor: the MLP input rows of the call from the C
    restatement, the regressor over ALL of them at once (its statistics are the call's), then the in_cube mask
    (lib/net/HGPIFuNet.py:274-275,361-363) -> (occ [N], X)"""
    _, X = query_icon(verts, faces, cmap, vis, feat, mlp.plain(), pts, sdf_clip=sdf_clip, calib=calib, cmap_local=cmap_local)
    pts = _f32(pts).reshape(-1, 3)
    xyz = np.empty_like(pts)
    lib().orc_project(_p(_calib12(calib)), _p(pts), C.c_int64(len(pts)), _p(xyz))
    in_cube = ((xyz > -1.0) & (xyz < 1.0)).all(1).astype(np.float32)
    return in_cube * mlp.forward(X), X


def query_icon_subset(verts, faces, cmap, vis, feat, mlp: Mlp, pts, subset, sdf_clip=0.05, calib=None, f64=False,
                      cmap_local=False):
    """query_icon over the call ``pts`` but occupancy only for ``pts[subset]`` (the geometry half and the
    outlier sign list still cover the whole call) -> (occ [M], X [M, C/2+7])"""
    verts, faces = _f32(verts).reshape(-1, 3), _i64(faces).reshape(-1, 3)
    cmap, vis, pts = _f32(cmap).reshape(-1, 3), _f32(vis).reshape(-1), _f32(pts).reshape(-1, 3)
    subset = _i64(subset).reshape(-1)
    feat = _f32(feat)
    feat = feat.reshape(feat.shape[-3:])
    Cc, H, W = feat.shape
    occ = np.empty(len(subset), np.float32)
    X = np.empty((len(subset), int(lib().orc_icon_c0(C.c_int(Cc)))), np.float32)
    cal = _calib12(calib)
    lib().orc_query_icon_subset(_p(verts), C.c_int64(len(verts)), _p(faces), C.c_int64(len(faces)), _p(cmap), _p(vis),
                                _p(feat), C.c_int(Cc), C.c_int(H), C.c_int(W), C.byref(mlp.struct),
                                C.c_float(np.float32(sdf_clip)), _p(cal), _p(pts), C.c_int64(len(pts)), _p(subset),
                                C.c_int64(len(subset)), _p(occ), _p(X), C.c_int(int(f64)), C.c_int(int(cmap_local)))
    return occ, X


def semantic_voxelize(verts, n_surface, code, tets, res=128, sigma=0.05, return_occ=False):
    """PaMIR semantic volume (restated voxelize_cuda.forward_semantic_voxelization, PARITY UNPINNED - see
    icon_accel.c): verts [V,3] (surface vertices first), code [n_surface,3], tets [T,4] -> [res,res,res,3] (z,y,x,c)"""
    verts, code, tets = _f32(verts).reshape(-1, 3), _f32(code).reshape(-1, 3), _i64(tets).reshape(-1, 4)
    assert len(code) == n_surface <= len(verts)
    out = np.empty((res, res, res, 3), np.float32)
    occ = np.empty((res, res, res), np.uint8)
    lib().orc_semantic_voxelize(_p(verts), C.c_int64(len(verts)), C.c_int64(n_surface), _p(code), _p(tets),
                                C.c_int64(len(tets)), C.c_int(res), C.c_float(np.float32(sigma)), _p(out), _p(occ))
    return (out, occ.astype(bool)) if return_occ else out


def query_vol(feat, vol, mlp: Mlp, pts, calib=None, f64=False):
    """PaMIR (vol [Cv,D,H,W]) or PIFu (vol None) branch -> (occ [N], X [N, c0])"""
    feat = _f32(feat)
    feat = feat.reshape(feat.shape[-3:])
    Cc, H, W = feat.shape
    pts = _f32(pts).reshape(-1, 3)
    n = len(pts)
    if vol is not None:
        vol = _f32(vol)
        vol = vol.reshape(vol.shape[-4:])
        Cv, Dv, Hv, Wv = vol.shape
        vp = _p(vol)
    else:
        Cv, Dv, Hv, Wv, vp = 1, 0, 0, 0, None
    occ = np.empty(n, np.float32)
    X = np.empty((n, Cc + Cv), np.float32)
    cal = _calib12(calib)
    lib().orc_query_vol(_p(feat), C.c_int(Cc), C.c_int(H), C.c_int(W), vp, C.c_int(Cv), C.c_int(Dv),
                        C.c_int(Hv), C.c_int(Wv), C.byref(mlp.struct), _p(cal), _p(pts), C.c_int64(n),
                        _p(occ), _p(X), C.c_int(int(f64)))
    return occ, X


def visibility(xy, z, faces, image_size: int = 4096, return_faces: bool = False):
    """get_visibility (lib/dataset/mesh_util.py:280-316): xy [V,2], z [V] or [V,1], faces [F,3] ->
    vis [V,1] float32 in {0,1} (and the winning face per pixel of the [0,1]^2 quadrant)."""
    xy, z, faces = _f32(xy).reshape(-1, 2), _f32(z).reshape(-1), _i64(faces).reshape(-1, 3)
    vis = np.empty(len(xy), np.float32)
    h = image_size // 2
    pf = np.empty((h, h), np.int64) if return_faces else None
    lib().orc_visibility(_p(xy), _p(z), C.c_int64(len(xy)), _p(faces), C.c_int64(len(faces)), C.c_int(image_size),
                         _p(vis), _p(pf) if return_faces else None)
    return (vis[:, None], pf) if return_faces else vis[:, None]
