"""Loader for libvo_hip.so (the C ABI declared in include/vo_hip.h).

The product has no CPU fallback: if the HIP library is missing, or no HIP device is usable,
loading / context creation raises.  Nothing here imports oracle/.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VO_HIP_LIB: another build of the same library (instrumented variants under tools/micro/)
LIB_PATH = os.environ.get("VO_HIP_LIB") or os.path.join(_HERE, "libvo_hip.so")
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/vo_hip.h declares
SYMBOLS = [
    "vo_limits_default", "vo_ctx_create", "vo_ctx_destroy", "vo_last_error", "vo_ctx_synchronize",
    "vo_orb_params_default", "vo_fast_detect", "vo_orb_detect_compute", "vo_orb_compute",
    "vo_knn2_hamming", "vo_match_filter", "vo_knn2_hamming_cross", "vo_match_filter_cross", "vo_knn2_hamming_static", "vo_find_essential", "vo_recover_pose", "vo_triangulate", "vo_five_point", "vo_essential_pose", "vo_akaze_params_default", "vo_akaze_detect_compute", "vo_debug_akaze_level",
    "vo_projection_matrix", "vo_pair_params_default", "vo_pairs_run_dev", "vo_pairs_get_frame",
    "vo_pairs_get_matches", "vo_prof_enable", "vo_prof_get", "vo_ctx_set_option", "vo_ctx_get_option", "vo_debug_read_level",
    "vo_lk_params_default", "vo_lk_track", "vo_lk_track_batch", "vo_essential_pose_batch", "vo_host_alloc_pinned", "vo_host_free_pinned", "vo_debug_read_lk_level",
    "vo_seq_create", "vo_seq_destroy", "vo_seq_last_error", "vo_seq_streams", "vo_seq_ctx", "vo_seq_run", "vo_seq_run_ex", "vo_seq_run_multi", "vo_seq_get_stats",
]


class AkazeParams(C.Structure):
    _fields_ = [("threshold", C.c_float), ("n_octaves", C.c_int32), ("n_sublevels", C.c_int32),
                ("channels", C.c_int32), ("pattern_size", C.c_int32), ("soffset", C.c_float),
                ("derivative_factor", C.c_float), ("kcontrast_percentile", C.c_float),
                ("kcontrast_nbins", C.c_int32), ("reserved", C.c_int32)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
                ("score_type", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32)]


class LkParams(C.Structure):
    _fields_ = [("win_w", C.c_int32), ("win_h", C.c_int32), ("max_level", C.c_int32), ("max_count", C.c_int32),
                ("epsilon", C.c_double), ("min_eig_threshold", C.c_double), ("channels", C.c_int32),
                ("reserved", C.c_int32)]


class Limits(C.Structure):
    _fields_ = [("max_width", C.c_int32), ("max_height", C.c_int32), ("max_features", C.c_int32),
                ("max_points", C.c_int32), ("max_batch", C.c_int32)]


class PairParams(C.Structure):
    _fields_ = [("orb", OrbParams), ("ratio", C.c_float), ("max_disp", C.c_float), ("prob", C.c_double),
                ("threshold", C.c_double), ("dist_thresh", C.c_double), ("seed", C.c_uint64),
                ("max_iters", C.c_int32), ("cross_check", C.c_int32)]


class SeqOutputs(C.Structure):
    _fields_ = [("matches", C.c_void_p), ("inlier_mask", C.c_void_p), ("pose_mask", C.c_void_p),
                ("match_cap", C.c_int32), ("reserved", C.c_int32)]


class PairResult(C.Structure):
    _fields_ = [("n_kp0", C.c_int32), ("n_kp1", C.c_int32), ("n_good", C.c_int32), ("n_inliers", C.c_int32),
                ("n_pose", C.c_int32), ("status", C.c_int32), ("E", C.c_double * 9), ("R", C.c_double * 9),
                ("t", C.c_double * 3)]


def build(force=False):
    """Compile the HIP library for gfx950 with the recipe in csrc/Makefile (cross-compiles)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libvo_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C visual_odometry_amd/csrc`); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.vo_last_error.restype = C.c_char_p
        _lib.vo_last_error.argtypes = [C.c_void_p]
        _lib.vo_ctx_destroy.argtypes = [C.c_void_p]
        _lib.vo_ctx_destroy.restype = None
        _lib.vo_seq_last_error.restype = C.c_char_p
        _lib.vo_seq_last_error.argtypes = [C.c_void_p]
        _lib.vo_seq_destroy.argtypes = [C.c_void_p]
        _lib.vo_seq_destroy.restype = None
        _lib.vo_seq_ctx.restype = C.c_void_p
        _lib.vo_seq_ctx.argtypes = [C.c_void_p, C.c_int]
        _lib.vo_seq_streams.argtypes = [C.c_void_p]
    return _lib
