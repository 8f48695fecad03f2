"""ctypes mirror of include/selfocc_hip.h (the C ABI of libselfocc_hip.so).

Field order and types must match the header exactly; tests/test_abi.py checks
``sizeof`` of every struct against the sizes the compiled library reports through
its exported symbols being callable with these layouts.
"""
import ctypes as C

ABI_VERSION = 32

# enums ---------------------------------------------------------------------------
RAYS_EXPLICIT, RAYS_PIXEL_GRID = 0, 1
SAMPLE_AT_START, SAMPLE_AT_MID = 0, 1
BKGD_NONE, BKGD_CONST, BKGD_PER_RAY = 0, 1, 2
JITTER_NONE, JITTER_SINGLE, JITTER_PER_BIN = 0, 1, 2
FLAG_DEPTH_DIV_NORM, FLAG_CLAMP_RGB, FLAG_EXACT, FLAG_NO_SKIP, FLAG_NO_FACE_SAFE, FLAG_NO_AHEAD, FLAG_RAY_PER_LANE = 1, 2, 4, 8, 16, 32, 64
VALUE_PIXEL_MAJOR, VALUE_HEAD_MAJOR = 0, 1
DTYPE_F32, DTYPE_BF16 = 0, 1
LINEAR_RELU = 1

_f, _i, _p = C.c_float, C.c_int32, C.c_void_p


class SoAxis(C.Structure):
    _fields_ = [("size0", _f), ("size1", _f), ("range0", _f), ("range1", _f),
                ("off0", _f), ("off1", _f), ("start", _f), ("tot_len", _i)]


class SoMapping(C.Structure):
    _fields_ = [("h", SoAxis), ("w", SoAxis), ("d", SoAxis)]


class SoRenderArgs(C.Structure):
    _fields_ = [
        ("map", SoMapping),
        ("sdf_vol", _p), ("feat_vol", _p),
        ("feat_dtype", _i), ("feat_stride", _i), ("n_rgb", _i), ("n_sem", _i),
        ("ray_mode", _i), ("n_rays", _i),
        ("origins", _p), ("dirs", _p), ("dir_norm", _p), ("img2lidar", _p),
        ("n_cams", _i), ("nx", _i), ("ny", _i),
        ("sx", _f), ("sy", _f), ("ox", _f), ("oy", _f),
        ("aabb", _f * 6), ("near_plane", _f),
        ("n_samples", _i), ("sample_pos", _i), ("jitter_mode", _i),
        ("t_rand", _p),
        ("inv_s", _f),
        ("bkgd_mode", _i), ("bkgd", _f * 3),
        ("bkgd_rays", _p),
        ("flags", _i),
        ("depth", _p), ("acc", _p), ("rgb", _p), ("sem", _p), ("max_depth", _p),
        ("nears", _p), ("fars", _p),
        ("weights", _p), ("ts", _p), ("deltas", _p), ("sdf", _p), ("grad", _p),
        ("sdf_brick", _p),
        ("inv_s_dev", _p),
    ]


class SoRenderBwdArgs(C.Structure):
    _fields_ = [
        ("fwd", SoRenderArgs),
        ("g_depth", _p), ("g_acc", _p), ("g_rgb", _p), ("g_sem", _p),
        ("g_weights", _p), ("g_sdf", _p), ("g_grad", _p),
        ("g_sdf_vol", _p), ("g_feat_vol", _p), ("g_inv_s", _p),
        ("scatter_ws", _p), ("scatter_ws_bytes", C.c_uint64),
    ]


class SoQueryArgs(C.Structure):
    _fields_ = [
        ("map", SoMapping),
        ("sdf_vol", _p), ("feat_vol", _p),
        ("feat_dtype", _i), ("feat_stride", _i), ("n_rgb", _i), ("n_sem", _i),
        ("xyz", _p), ("n", _i),
        ("sdf", _p), ("sem_logits", _p), ("sem_argmax", _p),
    ]


class SoOccArgs(C.Structure):
    _fields_ = [
        ("grid", _p), ("logits", _p),
        ("H", _i), ("W", _i), ("D", _i), ("C", _i),
        ("coords", _p),
        ("n0", _i), ("n1", _i), ("n2", _i),
        ("crop", _i * 6),
        ("thresh", _f), ("density", _i),
        ("lut", _p),
        ("sampled", _p), ("occ", _p), ("sem", _p),
    ]


class SoReprojArgs(C.Structure):
    _fields_ = [
        ("weights", _p), ("ts", _p), ("deltas", _p),
        ("pix", _p), ("curr_rgb", _p),
        ("T_prev", _p), ("T_next", _p),
        ("img_prev", _p), ("img_next", _p),
        ("R", _i), ("S", _i), ("Hi", _i), ("Wi", _i),
        ("img_h", _f), ("img_w", _f),
        ("l1", _p), ("rgb_combine", _p), ("any_valid", _p),
        ("wnorm", _p),
    ]


# every symbol include/selfocc_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "selfocc_abi_version": (C.c_int, []),
    "selfocc_last_error": (C.c_char_p, []),
    "selfocc_render_fwd": (C.c_int, [C.POINTER(SoRenderArgs), _p]),
    "selfocc_render_bwd": (C.c_int, [C.POINTER(SoRenderBwdArgs), _p]),
    "selfocc_render_bwd_ws_bytes": (C.c_size_t, [C.POINTER(SoRenderBwdArgs)]),
    "selfocc_msda_fwd": (C.c_int, [_p, _p, _p, _p, _p, _p] + [_i] * 7 + [_p]),
    "selfocc_msda_fused_fwd": (C.c_int, [_p, _p, _p, _p, _i, _p, _p, _p] + [_i] * 10 + [_p]),
    "selfocc_msda_cross_fwd": (C.c_int, [_p] * 8 + [_i] * 11 + [_p]),
    "selfocc_msda_cross_bwd": (C.c_int, [_p] * 12 + [_i] * 11 + [_p, C.c_size_t, _p]),
    "selfocc_msda_bwd": (C.c_int, [_p] * 9 + [_i] * 7 + [_p]),
    "selfocc_msda_banded_supported": (C.c_int, [_p] + [_i] * 6),
    "selfocc_msda_fused_bwd": (C.c_int, [_p] * 5 + [_i] + [_p] * 6 + [_i] * 11 + [_p, C.c_size_t, _p]),
    "selfocc_msda_bwd_banded_workspace": (C.c_size_t, [_i] * 5),
    "selfocc_msda_bwd_banded": (C.c_int, [_p] * 10 + [_i] * 7 + [_p, C.c_size_t, _p]),
    "selfocc_field_query": (C.c_int, [C.POINTER(SoQueryArgs), _p]),
    "selfocc_field_query_bwd": (C.c_int, [C.POINTER(SoQueryArgs), _p, _p, _p, _p, _p]),
    "selfocc_field_volume_bwd": (C.c_int, [_p] * 3 + [_i] * 4 + [_p, _p, _p, _i, _p, _p, _i] + [_p] * 7 + [_p]),
    "selfocc_field_volume_fwd": (C.c_int, [_p] * 3 + [_i] * 4 + [_p, _p, _i, _p, _p, _i, _p, _p, _i, _i, _p]),
    "selfocc_occ_resample": (C.c_int, [C.POINTER(SoOccArgs), _p]),
    "selfocc_iou_counts": (C.c_int, [_p, _p, _p, C.c_int64, _p, _i, _i, _p, _p]),
    "selfocc_layernorm_fwd": (C.c_int, [_p] * 6 + [C.c_int64, _i, C.c_float, _p]),
    "selfocc_layernorm_bwd_workspace": (C.c_size_t, [C.c_int64, _i]),
    "selfocc_layernorm_bwd": (C.c_int, [_p] * 8 + [C.c_int64, _i, _p, C.c_size_t, _p]),
    "selfocc_flatten_feats": (C.c_int, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "selfocc_point_sampling": (C.c_int, [_p] * 7 + [_i] * 4 + [C.c_float, C.c_float, _p]),
    "selfocc_linear_wgrad_supported": (C.c_int, [C.c_int64, _i, _i]),
    "selfocc_linear_wgrad_workspace": (C.c_size_t, [C.c_int64, _i, _i]),
    "selfocc_linear_wgrad": (C.c_int, [_p] * 4 + [C.c_int64, _i, _i, _p, C.c_size_t, _p]),
    "selfocc_linear_fwd_supported": (C.c_int, [C.c_int64, _i, _i]),
    "selfocc_linear_fwd": (C.c_int, [_p] * 4 + [_i, _p, _p, C.c_float, _p, _i, _p, _p, _p, C.c_int64, _i, _i, C.c_uint32, _p]),
    "selfocc_linear_fwd_heads": (C.c_int, [_p] * 4 + [C.c_int64, _i, _i, _i, C.c_uint32, _p]),
    "selfocc_linear_dgrad_supported": (C.c_int, [C.c_int64, _i, _i]),
    "selfocc_linear_dgrad_workspace": (C.c_size_t, [_i, _i]),
    "selfocc_linear_dgrad": (C.c_int, [_p] * 3 + [C.c_int64, _i, _i, _p, C.c_int64, _p]),
    "selfocc_second_diff_size": (C.c_size_t, [_i, _i, _i]),
    "selfocc_second_diff_fwd": (C.c_int, [_p, _p, _i, _i, _i, _p]),
    "selfocc_second_diff_bwd": (C.c_int, [_p, _p, _i, _i, _i, _p]),
    "selfocc_eikonal_partials": (C.c_int, [C.c_int64]),
    "selfocc_eikonal_fwd": (C.c_int, [_p, _p, C.c_int64, _p]),
    "selfocc_eikonal_bwd": (C.c_int, [_p, _p, _p, C.c_int64, _p]),
    "selfocc_dropout_add_fwd": (C.c_int, [_p, _p, _p, C.c_int64, C.c_float, C.c_uint64, _p]),
    "selfocc_dropout_bwd": (C.c_int, [_p, _p, C.c_int64, C.c_float, C.c_uint64, _p]),
    "selfocc_ssim_fwd": (C.c_int, [_p] * 4 + [_i] * 4 + [_p, _p]),
    "selfocc_ssim_bwd": (C.c_int, [_p] * 4 + [_i] * 4 + [_p, _p, _p, _p]),
    "selfocc_reproj_fwd": (C.c_int, [C.POINTER(SoReprojArgs), _p]),
    "selfocc_reproj_bwd": (C.c_int, [C.POINTER(SoReprojArgs), _p, _p, _p, _p]),
}
