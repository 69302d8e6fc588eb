"""The C ABI of libsvoxt_hip.so (include/svoxt.h) as ctypes sees it: the library, the five structs, every export with
its signature, the version check.  Nothing here knows about torch tensors (svox_t_amd/csrc/_marshal.py does)."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVOXT_LIB: load another build of the same ABI (kernel experiments); default in-tree.
LIB_PATH = os.environ.get("SVOXT_LIB") or os.path.join(_HERE, "libsvoxt_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension is not built. "
        "Run `python svox_t_amd/build.py` (needs hipcc; cross-compiles for gfx950).")

_lib = ctypes.CDLL(LIB_PATH)

ABI_VERSION = 22
FORMAT_RGBA, FORMAT_SH, FORMAT_SG, FORMAT_ASG = 0, 1, 2, 3


class _CTree(ctypes.Structure):          # struct svoxt_tree
    _fields_ = [
        ("features", ctypes.c_void_p), ("M", ctypes.c_int64),
        ("K", ctypes.c_int32), ("N", ctypes.c_int32),
        ("data", ctypes.c_void_p), ("child", ctypes.c_void_p),
        ("n_internal", ctypes.c_int64),
        ("offset", ctypes.c_void_p), ("scaling", ctypes.c_void_p),
        ("extra_data", ctypes.c_void_p),
        ("extra_rows", ctypes.c_int32), ("extra_cols", ctypes.c_int32),
        ("weight_accum", ctypes.c_void_p), ("xform", ctypes.c_void_p),
        ("accel", ctypes.c_void_p), ("accel_log2", ctypes.c_int32),
        ("xform_dim", ctypes.c_int32),
        ("sigma_mask", ctypes.c_void_p), ("sigma_mask_thresh", ctypes.c_float), ("reserved0", ctypes.c_int32),
        ("exp_table", ctypes.c_void_p),
    ]


class _CRays(ctypes.Structure):          # struct svoxt_rays
    _fields_ = [("origins", ctypes.c_void_p), ("dirs", ctypes.c_void_p),
                ("vdirs", ctypes.c_void_p), ("Q", ctypes.c_int64),
                ("image_width", ctypes.c_int32), ("image_height", ctypes.c_int32),
                ("c2w", ctypes.c_void_p), ("fx", ctypes.c_float), ("fy", ctypes.c_float),
                ("order", ctypes.c_void_p)]


class _CMotion(ctypes.Structure):        # struct svoxt_motion
    _fields_ = [("joint_features", ctypes.c_void_p), ("n_joints", ctypes.c_int32),
                ("feature_dim", ctypes.c_int32), ("skinning_weights", ctypes.c_void_p),
                ("joint_index", ctypes.c_void_p), ("n_bind", ctypes.c_int32)]


class _COptions(ctypes.Structure):       # struct svoxt_options
    _fields_ = [
        ("step_size", ctypes.c_float), ("background_brightness", ctypes.c_float),
        ("format", ctypes.c_int32), ("basis_dim", ctypes.c_int32),
        ("ndc_width", ctypes.c_int32), ("ndc_height", ctypes.c_int32),
        ("ndc_focal", ctypes.c_float),
        ("min_comp", ctypes.c_int32), ("max_comp", ctypes.c_int32),
        ("sigma_thresh", ctypes.c_float), ("stop_thresh", ctypes.c_float),
    ]


class _CLists(ctypes.Structure):         # struct svoxt_sample_lists
    _fields_ = [("rec", ctypes.c_void_p), ("aux", ctypes.c_void_p), ("max_samples", ctypes.c_int32),
                ("coef", ctypes.c_void_p), ("coef_bytes", ctypes.c_int64),
                ("terms", ctypes.c_void_p), ("terms_bytes", ctypes.c_int64),
                ("blocktab", ctypes.c_void_p), ("pool_blocks", ctypes.c_int64), ("pool_next", ctypes.c_void_p),
                ("terms_state", ctypes.c_int32), ("flags", ctypes.c_int32), ("tile_state", ctypes.c_void_p)]


class _CStep(ctypes.Structure):          # struct svoxt_step
    _fields_ = [("lists", _CLists), ("workspace_bytes", ctypes.c_int64), ("records", ctypes.c_int32),
                ("grad_cols", ctypes.c_int32), ("grad_stride", ctypes.c_int32), ("uses_mask", ctypes.c_int32),
                ("uses_table", ctypes.c_int32)] + \
               [(n, ctypes.c_int64) for n in ("off_mask", "off_table", "off_tables", "tables_bytes", "off_rec", "off_aux",
                                              "off_terms", "terms_bytes", "off_grad_rows", "off_bwd_ws", "bwd_ws_bytes", "nt")] + \
               [(n, ctypes.c_int32) for n in ("pad_K", "pad_real", "pad_dummy", "pad_w")] + \
               [(n, ctypes.c_int64) for n in ("off_pad_features", "off_pad_out", "off_pad_gout", "off_pad_grad")]


# svoxt_sample_lists.flags (include/svoxt.h)
LISTS_NATIVE_MATH, LISTS_FWD_ONE_KERNEL, LISTS_FWD_TWO_KERNELS, LISTS_FWD_NO_OVERLAP, LISTS_GRAD_ZEROED, LISTS_BEGUN = 1, 2, 4, 8, 16, 32
LISTS_TEST_DROP, LISTS_TEST_NOPOLL, LISTS_TEST_STALE, LISTS_FWD_AGENT_FENCE = 256, 512, 1024, 2048


_P = ctypes.POINTER
_vp, _i32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64

# Every symbol include/svoxt.h declares, with its signature.
EXPORTS = {
    "svoxt_abi_version": (ctypes.c_int, []),
    "svoxt_last_error": (ctypes.c_char_p, []),
    "svoxt_out_data_dim": (ctypes.c_int, [_P(_COptions), _i32]),
    "svoxt_volume_render_fwd": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp]),
    "svoxt_fwd_workspace_bytes": (ctypes.c_int64, [_i64, _i32]),
    "svoxt_volume_render_fwd_ws": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp, _i64, _i32, _vp]),
    "svoxt_volume_render_fwd_scratch": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _P(_CLists), _i32, _vp]),
    "svoxt_volume_render_bwd": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _i32, _vp, _i32, _vp, _i64, _vp]),
    "svoxt_bwd_workspace_bytes": (ctypes.c_int64, [_i64, _i32]),
    "svoxt_can_record": (ctypes.c_int, [_P(_CTree), _P(_COptions)]),
    "svoxt_fwd_fills_terms": (ctypes.c_int, [_P(_CTree), _P(_COptions), _i32]),
    "svoxt_sigma_mask_bytes": (ctypes.c_int64, [ctypes.c_int64]),
    "svoxt_sigma_mask_build": (ctypes.c_int, [_P(_CTree), ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "svoxt_sigma_mask_build_fill": (ctypes.c_int, [_P(_CTree), ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, _i64,
                                                   ctypes.c_void_p]),
    "svoxt_exp_table_build": (ctypes.c_int, [_P(_CTree), ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "svoxt_compact_rows": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "svoxt_compact_rows_clear": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "svoxt_query_leaves_workspace_bytes": (ctypes.c_int64, [_i64]),
    "svoxt_query_leaves": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "svoxt_volume_render_fwd_record": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _P(_CLists), _vp]),
    "svoxt_volume_render_bwd_replay": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _i32, _vp, _i32, _P(_CLists), _vp, _vp]),
    "svoxt_opacity_render_fwd": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp]),
    "svoxt_opacity_render_bwd": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp, _vp]),
    "svoxt_opacity_render_fwd_record": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _P(_CLists), _vp]),
    "svoxt_opacity_render_bwd_replay": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp,
                                                        ctypes.c_int32, _P(_CLists), _vp]),
    "svoxt_render_depth": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp]),
    "svoxt_query_fwd": (ctypes.c_int, [_P(_CTree), _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "svoxt_query_bwd": (ctypes.c_int, [_P(_CTree), _vp, _i64, _vp, _vp, _vp]),
    "svoxt_count_fwd": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp]),
    "svoxt_count_touched": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp, _vp, _vp]),
    "svoxt_set_bwd_counters": (ctypes.c_int, [_vp]),
    "svoxt_set_bwd_check": (ctypes.c_int, [_vp]),
    "svoxt_set_super_tile_rows": (_i64, [_i64]),
    "svoxt_image_walk": (_i32, [_P(_CTree), _P(_CRays)]),
    "svoxt_step_plan": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _i64, _P(_CStep)]),
    "svoxt_step_forward": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _P(_CStep), _vp, _vp]),
    "svoxt_step_backward": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp, _P(_CStep), _vp, _vp]),
    "svoxt_ray_order_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64]),
    "svoxt_ray_order": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp, ctypes.c_int64, _vp]),
    "svoxt_image_probe": (ctypes.c_int, [_P(_CRays), _vp, _i32, _vp]),
    "svoxt_gather_rays": (ctypes.c_int, [_P(_CRays), _vp, _vp, _vp, _vp, _vp]),
    "svoxt_permute_rows": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _vp]),
    "svoxt_accel_bytes": (ctypes.c_int64, [_i32, ctypes.c_int64]),
    "svoxt_accel_build": (ctypes.c_int, [_P(_CTree), _i32, _vp, _vp]),
    "svoxt_build_workspace_bytes": (ctypes.c_int64, [_i32]),
    "svoxt_build_count": (ctypes.c_int, [_vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _vp]),
    "svoxt_build_emit": (ctypes.c_int, [_vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "svoxt_construct_tree": (ctypes.c_int, [_P(_CTree), _vp, _i64, _vp]),
    "svoxt_refine": (ctypes.c_int, [_vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "svoxt_motion_render": (ctypes.c_int, [_P(_CTree), _P(_CRays), _P(_COptions), _vp, _vp, _vp, _vp, _vp]),
    "svoxt_warp_vertices": (ctypes.c_int, [_vp, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp]),
    "svoxt_warp_vertices_bwd": (ctypes.c_int, [_vp, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "svoxt_motion_workspace_bytes": (ctypes.c_int64, [_i64, _i32]),
    "svoxt_motion_feature_render_fwd": (ctypes.c_int, [_P(_CTree), _P(_CMotion), _P(_CRays), _P(_COptions), _vp, _vp, _i64, _vp]),
    "svoxt_motion_feature_render_bwd": (ctypes.c_int, [_P(_CTree), _P(_CMotion), _P(_CRays), _P(_COptions), _vp, _vp, _vp, _i64, _vp]),
}
for _name, (_res, _args) in EXPORTS.items():
    _fn = getattr(_lib, _name)       # AttributeError here = library/header mismatch
    _fn.restype = _res
    _fn.argtypes = _args

if _lib.svoxt_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH}: ABI version {_lib.svoxt_abi_version()} != {ABI_VERSION}; rebuild")
