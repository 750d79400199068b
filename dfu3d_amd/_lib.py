"""ctypes binding of libdfu3d_hip.so (C ABI in include/dfu3d.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, every
entry point raises.  Signatures are declared from the header so a mismatch is
caught at load time (tests/test_abi.py checks every symbol of include/dfu3d.h).
"""
import ctypes
import os
import re

from . import _build

c_void_p = ctypes.c_void_p
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64
c_f64 = ctypes.c_double

HEADER = os.path.join(_build.INCLUDE, "dfu3d.h")


class BinGeom(ctypes.Structure):
    """dfu3d_bin_geom (include/dfu3d.h)."""
    _fields_ = [
        ("vsize_r", c_f64), ("vsize_t", c_f64), ("vsize_p", c_f64),
        ("rmin_r", c_f64), ("rmin_t", c_f64), ("rmin_p", c_f64),
        ("grid_r", c_i32), ("grid_t", c_i32), ("grid_p", c_i32),
        ("t_lo", c_i32), ("t_n", c_i32), ("p_lo", c_i32), ("p_n", c_i32),
        ("max_points_per_voxel", c_i32), ("max_voxels", c_i32),
        ("theta_min", c_f64), ("z_max", c_f64), ("depth_min", c_f64),
    ]


class Sizes(ctypes.Structure):
    """dfu3d_sizes (include/dfu3d.h)."""
    _fields_ = [
        ("V", c_i32), ("H", c_i32), ("W", c_i32), ("max_inst", c_i32),
        ("cap_n", c_i32), ("cap_vox", c_i32), ("cap_rows", c_i32), ("max_points_per_voxel", c_i32),
        ("pool_cap", c_i64), ("table_entries", c_i64), ("dense", c_i32), ("stat_filter", c_i32),
    ]


class ChainCfg(ctypes.Structure):
    """dfu3d_chain_cfg (include/dfu3d.h)."""
    _fields_ = [
        ("V", c_i32), ("H", c_i32), ("W", c_i32), ("max_inst", c_i32), ("cap_n", c_i32), ("cap_vox", c_i32),
        ("cap_rows", c_i32),
        ("dense", c_i32), ("apply_fov", c_i32), ("fov_h", c_i32), ("fov_w", c_i32), ("stat_filter", c_i32),
        ("bounds_h", c_i32), ("bounds_w", c_i32), ("mask_format", c_i32), ("reserved0", c_i32),
        ("stat_voxel", c_f64),
        ("pool_cap", c_i64),
        ("plane_max_hs", c_f64), ("plane_range", c_f64), ("plane_offset", c_f64),
        ("ransac_trials", c_i32), ("nb_points", c_i32),
        ("ransac_seed", c_u64),
        ("fuse_C", c_f64), ("R0", c_f64), ("Rd", c_f64),
        ("n_theta", c_i32), ("stat_nb_neighbors", c_i32),
        ("dtheta", c_f64), ("car_aspect_max", c_f64), ("stat_std_ratio", c_f64),
        ("geom", BinGeom),
    ]


class EvalCombo(ctypes.Structure):
    """dfu3d_eval_combo (include/dfu3d.h)."""
    _fields_ = [("cls", c_i32), ("difficulty", c_i32), ("min_overlap", c_f64)]


_P = c_void_p
# argument lists mirror include/dfu3d.h exactly (pointers as void*)
SIGNATURES = {
    "dfu3d_version": (c_i32, []),
    "dfu3d_strerror": (ctypes.c_char_p, [c_i32]),
    "dfu3d_bin_table_geometry": (c_i64, [ctypes.POINTER(BinGeom)]),
    "dfu3d_fov_filter": (c_i32, [_P, _P, _P, _P, c_i32, c_i32, c_i32, c_i32, _P, _P, _P]),
    "dfu3d_plane_ransac": (c_i32, [_P, _P, _P, _P, _P, c_i32, c_i32, c_f64, c_f64, c_i32,
                                   c_u64, _P, _P, _P, _P]),
    "dfu3d_pack_masks": (c_i32, [_P, _P, c_i32, c_i32, c_i32, c_i32, _P, c_i32, _P]),
    "dfu3d_project_label": (c_i32, [_P, _P, _P, _P, _P, _P, _P, _P, c_i32, _P, c_i32, c_i32, c_i32,
                                    c_i32, c_i32, c_i32, c_i32, c_f64, c_f64, _P, _P, _P, _P, _P, _P, _P,
                                    _P, _P]),
    "dfu3d_bin_table_init": (c_i32, [_P, c_i64, _P]),
    "dfu3d_backproject_scratch_words": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i64,
                                                ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "dfu3d_backproject_bin": (c_i32, [_P, _P, _P, c_i32, _P, c_i32, c_i32, c_i32, c_i32,
                                      ctypes.POINTER(BinGeom), c_i32, _P, _P, _P, c_i32, _P,
                                      _P, _P, _P, _P, _P, _P, c_i32, _P]),
    "dfu3d_segments_build": (c_i32, [_P, _P, _P, _P, _P, c_i32, _P, _P, _P, _P, _P, c_i32,
                                     c_i32, c_i32, c_i64, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P, _P, _P, _P, _P, _P, _P]),
    "dfu3d_segments_scratch_words": (c_i64, [c_i32, c_i32, c_i32]),
    "dfu3d_rf_shadow_bytes": (c_i64, [c_i64]),
    "dfu3d_rf_queue_ints": (c_i64, [c_i64]),
    "dfu3d_radius_filter": (c_i32, [_P, _P, _P, _P, _P, _P, c_i32, c_i32, c_i64, _P, _P, _P, _P, _P,
                                    c_i32, _P]),
    "dfu3d_stat_filter": (c_i32, [_P, _P, _P, _P, _P, _P, c_i32, c_f64, c_i32, c_i64, _P, _P,
                                  _P, _P, _P]),
    "dfu3d_voxel_down_sample_scratch_bytes": (c_i64, [c_i64]),
    "dfu3d_voxel_down_sample": (c_i32, [_P, _P, _P, _P, _P, _P, c_f64, c_i32, c_i64, _P, _P, _P]),
    "dfu3d_ballquery_fuse": (c_i32, [_P, _P, _P, _P, _P, _P, _P, c_f64, c_i32, c_i64, _P, _P,
                                     _P]),
    "dfu3d_ballquery_fuse_masked": (c_i32, [_P, _P, _P, _P, _P, _P, _P, c_f64, c_i32, c_i64, _P, _P,
                                     _P]),
    "dfu3d_ballquery_fuse_joint": (c_i32, [_P, _P, _P, _P, _P, _P, _P, c_f64, c_i32, c_i64, _P, _P,
                                     _P]),
    "dfu3d_range_cluster": (c_i32, [_P, _P, _P, _P, c_i32, c_f64, c_f64, _P, _P, _P, _P, c_i64,
                                    _P]),
    "dfu3d_points_in_boxes_mask": (c_i32, [_P, c_i32, c_i32, _P, c_i32, _P, _P]),
    "dfu3d_gt_database": (c_i32, [_P, _P, _P, _P, c_i32, _P, _P, c_i64, _P, _P, _P, _P]),
    "dfu3d_boxes_bev": (c_i32, [_P, c_i32, _P, c_i32, _P, c_i32, _P]),
    "dfu3d_nms_bev": (c_i32, [_P, c_i32, ctypes.c_float, _P, _P, _P, _P]),
    "dfu3d_nms_normal_bev": (c_i32, [_P, c_i32, ctypes.c_float, _P, _P, _P, _P]),
    "dfu3d_boxes_bev_paired": (c_i32, [_P, _P, c_i32, _P, c_i32, _P]),
    "dfu3d_rotate_iou_eval": (c_i32, [_P, c_i32, _P, c_i32, _P, c_i32, _P]),
    "dfu3d_eval_overlaps": (c_i32, [c_i32, c_i32, _P, _P, _P, _P, _P, _P, _P, _P, c_i64, _P]),
    "dfu3d_eval_match_scores": (c_i32, [c_i32, c_i32, c_i32] + [_P] * 15 + [c_i32, c_i64, _P, _P, _P]),
    "dfu3d_eval_match_stats": (c_i32, [c_i32, c_i32, c_i32] + [_P] * 15 + [c_i32, _P, _P, c_i32, c_i32, _P, _P, _P]),
    "dfu3d_workspace_bytes": (c_i64, [c_i32, ctypes.POINTER(Sizes)]),
    "dfu3d_chain_workspace_bytes": (c_i64, [ctypes.POINTER(ChainCfg)]),
    "dfu3d_chain_workspace_init": (c_i32, [ctypes.POINTER(ChainCfg), _P, _P]),
    "dfu3d_pseudo_boxes": (c_i32, [ctypes.POINTER(ChainCfg)] + [_P] * 20),
    "dfu3d_la_sampling": (c_i32, [_P, c_i32, _P, c_i32, ctypes.c_float, ctypes.c_float, _P, _P, _P, c_i64, _P]),
    "dfu3d_selftest_classify": (c_i32, [_P, c_i32, c_i32, _P, c_i32, c_i64, ctypes.c_uint64, c_f64, c_f64, _P, _P, _P]),
    "dfu3d_selftest_backproject": (c_i32, [_P, c_i32, c_i32, c_i64, ctypes.c_uint64, c_f64, c_f64, _P, _P, _P]),
    "dfu3d_lshape_fit_ws_doubles": (c_i64, [c_i64, c_i32]),
    "dfu3d_lshape_fit": (c_i32, [_P, _P, _P, _P, _P, _P, c_i32, c_i32, _P, _P, _P, _P, _P,
                                 c_i32, c_f64, c_f64, _P, _P, _P, c_i32, _P, _P, _P, _P, c_i64,
                                 _P]),
}

_LIB = None


class Dfu3dError(RuntimeError):
    pass


def header_symbols():
    """Function names declared in include/dfu3d.h."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfu3d_[a-z0-9_]+)\s*\(", text)))


def header_version():
    """DFU3D_VERSION of include/dfu3d.h."""
    return int(re.search(r"#define\s+DFU3D_VERSION\s+(\d+)", open(HEADER).read()).group(1))


def lib():
    """Load (building in-tree if needed) libdfu3d_hip.so; raises if impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm ships its own HIP runtime; it must be in the process before this
    # library is dlopen'ed so that both resolve the SAME libamdhip64 (otherwise torch's
    # streams and device pointers are foreign to our launches: hipErrorInvalid*).
    import torch  # noqa: F401
    path = _build.OUT
    try:
        _build.build()          # no-op when the library is newer than every source
    except Exception as e:      # no silent fallback: a stale library must not be mistaken for the sources
        if not os.path.exists(path):
            raise Dfu3dError("libdfu3d_hip.so is missing and could not be built: %r" % (e,))
        if _build.needs_build():
            raise Dfu3dError("libdfu3d_hip.so is older than its sources and could not be rebuilt: %r" % (e,))
    try:
        L = ctypes.CDLL(path)
    except OSError as e:
        raise Dfu3dError("cannot load %s: %s" % (path, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise Dfu3dError("%s does not export %s" % (path, name))
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def load_variant(name):
    """A test / timing build of the library (dfu3d_amd/_build.py: VARIANTS) with the product's signatures.  For tests/ and
    tools/ only: lib() never returns one, whatever the environment says."""
    import torch  # noqa: F401
    path = _build.build_variant(name)        # no-op when newer than every source
    L = ctypes.CDLL(path)
    for sym, (res, args) in SIGNATURES.items():
        fn = getattr(L, sym)
        fn.restype = res
        fn.argtypes = args
    return L


def check(code, what):
    if code != 0:
        msg = lib().dfu3d_strerror(int(code)).decode()
        raise Dfu3dError("%s failed: %s (%d)" % (what, msg, code))
