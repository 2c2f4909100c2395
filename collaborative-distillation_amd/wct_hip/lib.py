"""ctypes binding of libwct_hip.so (the C ABI declared in include/wct_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# WCT_LIB_PATH: load an instrumented build of the same library (tools/experiments/sp_timing.sh); default = the in-tree build
LIB_PATH = os.environ.get("WCT_LIB_PATH") or os.path.join(_PKG, "libwct_hip.so")

WCT_OK, WCT_ERR_INVALID, WCT_ERR_HIP, WCT_ERR_NOMEM, WCT_ERR_STATE, WCT_ERR_RANGE = 0, -1, -2, -3, -4, -5
KIND_ENC, KIND_DEC = 0, 1
LAYOUT_NHWC, LAYOUT_NCHW = 0, 1

# every symbol include/wct_hip.h declares (tests check that the built library exports all of them)
SYMBOLS = [
    "wct_version", "wct_create", "wct_destroy", "wct_last_error", "wct_set_stream", "wct_sync", "wct_saturation_count", "wct_range_poll", "wct_range_flag_f64", "wct_debug_set", "wct_debug_get",
    "wct_load_module", "wct_feature_shape", "wct_encode", "wct_decode", "wct_moments", "wct_solve",
    "wct_apply", "wct_transform", "wct_decode_affine", "wct_style_transfer_level", "wct_stylize",
    "wct_style_prepare", "wct_content_encode", "wct_content_solve", "wct_content_decode",
    "wct_comm_load", "wct_comm_library", "wct_comm_unique_id", "wct_comm_init", "wct_comm_attach", "wct_comm_destroy", "wct_level_sharded",
    "wct_comm_attach_collectives", "wct_comm_info", "wct_comm_selftest", "wct_shard_geometry", "wct_stylize_sharded", "wct_style_moments", "wct_style_solve",
    "wct_style_prepare_levels", "wct_style_stats_count", "wct_style_export", "wct_style_import", "wct_stylize_prepared",
    "wct_u8_to_planar", "wct_planar_to_u8", "wct_stylize_u8", "wct_resize_shape", "wct_resize_u8", "wct_resize_u8_to_planar",
    "wct_workspace_bytes", "wct_reserve", "wct_set_conv_mode", "wct_set_numpy_variant", "wct_set_overlap", "wct_profile_enable", "wct_profile_reset", "wct_profile_read",
]


class WctLayer(ctypes.Structure):
    _fields_ = [("cin", c_int), ("cout", c_int), ("pool_after", c_int), ("up_after", c_int),
                ("weight", POINTER(c_float)), ("bias", POINTER(c_float))]


class WctProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("ms", c_double), ("flops", c_double), ("bytes", c_double),
                ("launches", c_long)]


class WctP2P(ctypes.Structure):
    _fields_ = [("peer", c_int), ("is_send", c_int), ("buf", c_void_p), ("bytes", c_size_t)]


ALL_REDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_void_p)          # (user, buf, count, stream)
BROADCAST_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)    # (user, buf, bytes, root, stream)
SENDRECV_FN = ctypes.CFUNCTYPE(c_int, c_void_p, POINTER(WctP2P), c_int, c_void_p)        # (user, ops, n_ops, stream)


class WctCollectives(ctypes.Structure):
    """include/wct_hip.h wct_collectives: the transport table of wct_stylize_sharded (wct_comm_attach_collectives)."""
    _fields_ = [("user", c_void_p), ("all_reduce_sum_f64", ALL_REDUCE_FN), ("broadcast", BROADCAST_FN), ("sendrecv", SENDRECV_FN)]


HALO_MODES = {"auto": 0, "recompute": 1, "exchange": 2}
STYLE_MODES = {"auto": 0, "owner": 1, "strips": 2, "replicate": 3}
SHARD_BROADCAST_MAP = 1
SHARD_FAST_FOLD = 2


class WctError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libwct_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load() -> ctypes.CDLL:
    """dlopen the in-tree library.  Raises ImportError when it has not been built
    (`python -c 'import __graft_entry__ as g; g.build()'` or collaborative-distillation_amd/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libwct_hip.so not found at %s -- build it with collaborative-distillation_amd/build.sh; "
                          "there is no CPU fallback" % LIB_PATH)
    # torch first: it brings its own HIP runtime, and both must resolve to ONE libamdhip64 in the process -- with this
    # library loaded before torch, wct_create later fails with a HIP error (seen when build() and smoke() share a process)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    fp, dp, ip = POINTER(c_float), POINTER(c_double), POINTER(c_int)
    vp = c_void_p  # device pointers travel as integers
    lib.wct_version.restype = c_int
    lib.wct_create.argtypes = [c_int, POINTER(c_void_p)]
    lib.wct_destroy.argtypes = [c_void_p]
    lib.wct_destroy.restype = None
    lib.wct_last_error.argtypes = [c_void_p]
    lib.wct_last_error.restype = c_char_p
    lib.wct_set_stream.argtypes = [c_void_p, c_void_p]
    lib.wct_sync.argtypes = [c_void_p]
    lib.wct_saturation_count.argtypes = [c_void_p, c_int, POINTER(ctypes.c_ulonglong)]
    lib.wct_range_poll.argtypes = [c_void_p, POINTER(ctypes.c_ulonglong)]
    lib.wct_range_flag_f64.argtypes = [c_void_p, vp]
    lib.wct_debug_get.argtypes = [c_void_p, c_char_p, POINTER(ctypes.c_double)]
    lib.wct_debug_set.argtypes = [c_void_p, c_char_p, c_double]
    lib.wct_load_module.argtypes = [c_void_p, c_int, c_int, c_int, POINTER(WctLayer), fp, fp]
    lib.wct_feature_shape.argtypes = [c_void_p, c_int, c_int, c_int, ip, ip, ip]
    lib.wct_encode.argtypes = [c_void_p, c_int, vp, c_int, c_int, vp, c_int]
    lib.wct_decode.argtypes = [c_void_p, c_int, vp, c_int, c_int, c_int, vp]
    lib.wct_moments.argtypes = [c_void_p, vp, c_int, c_int, c_int, c_int, c_int, vp, vp]
    lib.wct_solve.argtypes = [c_void_p, c_int, c_double, vp, vp, c_double, vp, vp, c_double, vp, vp, ip]
    lib.wct_apply.argtypes = [c_void_p, vp, c_int, c_int, c_int, c_int, vp, vp, vp]
    lib.wct_transform.argtypes = [c_void_p, vp, c_int, c_int, c_int, vp, c_int, c_int, c_float, c_int, vp]
    lib.wct_decode_affine.argtypes = [c_void_p, c_int, vp, c_int, c_int, vp, vp, vp]
    lib.wct_style_transfer_level.argtypes = [c_void_p, c_int, vp, c_int, c_int, vp, c_int, c_int, c_float, vp, ip, ip]
    lib.wct_stylize.argtypes = [c_void_p, vp, c_int, c_int, vp, c_int, c_int, c_float, c_int, vp, ip, ip]
    lib.wct_style_prepare.argtypes = [c_void_p, vp, c_int, c_int]
    lib.wct_content_encode.argtypes = [c_void_p, c_int, vp, c_int, c_int, c_int, c_int, vp, vp, ip, ip]
    lib.wct_content_solve.argtypes = [c_void_p, c_int, c_double, vp, vp, c_float, vp, vp]
    lib.wct_content_decode.argtypes = [c_void_p, c_int, vp, vp, vp, ip, ip]
    lib.wct_comm_load.argtypes = [c_char_p]
    lib.wct_comm_library.restype = c_char_p
    lib.wct_comm_attach_collectives.argtypes = [c_void_p, POINTER(WctCollectives), c_int, c_int]
    lib.wct_comm_info.argtypes = [c_void_p, ip, ip]
    lib.wct_comm_selftest.argtypes = [c_void_p]
    lib.wct_shard_geometry.argtypes = [c_int, c_int, c_int, c_int, ip, ip, ip, ip, ip]
    lib.wct_stylize_sharded.argtypes = [c_void_p, vp, c_int, c_int, c_int, c_int, vp, c_int, c_int, c_float, c_int, c_int, c_int, vp, ip, ip, vp]
    lib.wct_style_moments.argtypes = [c_void_p, c_int, vp, c_int, c_int, c_int, c_int, vp, vp]
    lib.wct_style_solve.argtypes = [c_void_p, c_int, c_double, vp, vp]
    lib.wct_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.wct_comm_init.argtypes = [c_void_p, c_int, c_int, ctypes.c_char_p]
    lib.wct_comm_attach.argtypes = [c_void_p, c_void_p, c_int, c_int]
    lib.wct_comm_destroy.argtypes = [c_void_p]
    lib.wct_level_sharded.argtypes = [c_void_p, c_int, vp, c_int, c_int, c_int, c_int, c_double, c_float, vp, ip, ip, vp]
    lib.wct_style_prepare_levels.argtypes = [c_void_p, vp, c_int, c_int, ctypes.c_uint]
    lib.wct_style_stats_count.argtypes = [c_void_p, c_int, POINTER(c_size_t)]
    lib.wct_style_export.argtypes = [c_void_p, c_int, vp]
    lib.wct_style_import.argtypes = [c_void_p, c_int, vp]
    lib.wct_stylize_prepared.argtypes = [c_void_p, vp, c_int, c_int, c_float, c_int, vp, ip, ip]
    lib.wct_u8_to_planar.argtypes = [c_void_p, vp, c_int, c_int, vp]
    lib.wct_planar_to_u8.argtypes = [c_void_p, vp, c_int, c_int, vp, c_int]
    lib.wct_stylize_u8.argtypes = [c_void_p, vp, c_int, c_int, vp, c_int, c_int, c_float, c_int, vp, ip, ip, c_int]
    lib.wct_resize_shape.argtypes = [c_int, c_int, c_int, ip, ip]
    lib.wct_resize_u8.argtypes = [c_void_p, vp, c_int, c_int, vp, c_int, c_int]
    lib.wct_resize_u8_to_planar.argtypes = [c_void_p, vp, c_int, c_int, vp, c_int, c_int]
    lib.wct_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
    lib.wct_workspace_bytes.restype = c_size_t
    lib.wct_reserve.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
    lib.wct_set_conv_mode.argtypes = [c_void_p, c_int]
    lib.wct_set_numpy_variant.argtypes = [c_void_p, c_int]
    lib.wct_set_overlap.argtypes = [c_void_p, c_int]
    lib.wct_profile_enable.argtypes = [c_void_p, c_int]
    lib.wct_profile_reset.argtypes = [c_void_p]
    lib.wct_profile_read.argtypes = [c_void_p, POINTER(WctProfEntry), c_int, ip]
    _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != WCT_OK:
        msg = lib.wct_last_error(ctx).decode("utf-8", "replace") if ctx else "no context"
        if rc == WCT_ERR_INVALID:
            raise ValueError("libwct_hip: " + msg)
        if rc == WCT_ERR_RANGE:
            raise OverflowError("libwct_hip: " + msg)
        raise WctError(rc, msg)
