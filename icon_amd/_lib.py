"""ctypes binding of libicon_amd.so (the C ABI declared in include/icon_amd.h).

The shared library is built in-tree by ``icon_amd/csrc/Makefile`` (hipcc, gfx950).  There is no
CPU fallback: if the library is missing, or no HIP device is visible, every compute entry point
raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICON_AMD_LIB") or os.path.join(_HERE, "libicon_amd.so")   # ICON_AMD_LIB: experiment builds (tools/exp_fused.sh)
CSRC = os.path.join(_HERE, "csrc")

# include/icon_amd.h enums
PRIOR = {"icon": 0, "pamir": 1, "pifu": 2}
CMAP = {"reference": 0, "local": 1}
SEARCH = {"bvh": 0, "brute": 1}
PRECISION = {"f32": 0, "f16x3": 1}

# every symbol include/icon_amd.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "icon_last_error", "icon_version", "icon_device_count",
    "icon_mesh_create", "icon_mesh_destroy", "icon_mesh_vertex_normals", "icon_mesh_stats",
    "icon_mesh_arena_bytes", "icon_mesh_create_arena", "icon_mesh_status", "icon_debug_set_mesh_build", "icon_debug_mesh_layout", "icon_debug_host_mesh_build",
    "icon_sdf_query",
    "icon_feat_create", "icon_feat_destroy", "icon_feat_set_smpl_feats",
    "icon_mlp_create", "icon_mlp_destroy", "icon_mlp_forward", "icon_mlp_set_last_op",
    "icon_work_create", "icon_work_destroy", "icon_work_profile", "icon_work_stage_ms", "icon_work_profile_detail", "icon_work_profile_workgroups", "icon_work_set_steal", "icon_work_set_reserve_cus",
    "icon_query_points", "icon_query_points_dcalib", "icon_query_rows", "icon_grid_rows",
    "icon_volume_any_above", "icon_adaptive_eval", "icon_adaptive_counts", "icon_adaptive_reruns", "icon_grid_eval_slab", "icon_grid_slab_features", "icon_grid_slab_finish", "icon_grid_slab_features_msg",
    "icon_grid_slab_finish_gathered", "icon_debug_set_shell_skip", "icon_debug_set_option", "icon_work_status", "icon_sdf_query_ties", "icon_work_set_tie_rule",
    "icon_export_mesh", "icon_mc_count", "icon_mc_emit", "icon_mc_count_range", "icon_mc_emit_keyed", "icon_debug_traversal_stats", "icon_debug_set_unfused",
    "icon_visibility", "icon_mesh_components", "icon_clean_mesh", "icon_semantic_voxelize",
]

_lib = None


class IconAmdError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile libicon_amd.so with hipcc for gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-j8", "-C", CSRC] + (["-B"] if force else [])
    proc = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise IconAmdError("building libicon_amd.so failed:\n" + proc.stdout[-4000:])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IconAmdError(f"{LIB_PATH} not found - run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C icon_amd/csrc` (needs hipcc); there is no CPU fallback")
    # torch first: its wheel bundles its own libamdhip64 and must be the HIP runtime of the process - if this library
    # (linked against /opt/rocm) pulled the system runtime in before `import torch`, torch would later report "No HIP GPUs
    # are available" (seen with build() followed by smoke() in one interpreter)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.icon_last_error.restype = C.c_char_p
    for name in SYMBOLS:
        if name != "icon_last_error":
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().icon_last_error().decode("utf-8", "replace")
        raise IconAmdError(f"{what or 'icon_amd'} failed (code {rc}): {msg}")


def device_count() -> int:
    return int(lib().icon_device_count())


def require_device() -> None:
    if device_count() < 1:
        raise IconAmdError("icon_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback")


def ptr(t) -> C.c_void_p:
    """Device (or host) address of a contiguous torch tensor / numpy array."""
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous(), "tensor must be contiguous"
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)
