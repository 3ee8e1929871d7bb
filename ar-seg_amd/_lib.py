"""ctypes binding of libarseg_hip.so (the C ABI declared in include/arseg_hip.h).

The library is the product: if it is missing or a symbol is absent this module raises -- there
is no Python / PyTorch fallback for any op.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int16, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARSEG_HIP_LIB", os.path.join(_HERE, "lib", "libarseg_hip.so"))   # env override: kernel experiments

ABI_VERSION = 5          # ARSEG_ABI_VERSION of include/arseg_hip.h
ARSEG_OK, ARSEG_EINVAL, ARSEG_EUNSUPPORTED, ARSEG_EWORKSPACE = 0, -1, -2, -3
ACT_NONE, ACT_RELU, ACT_PRELU, ACT_SIGMOID = 0, 1, 2, 3
NCHW, NHWC, C8 = 0, 1, 2
FLOW_F32, FLOW_F64 = 0, 1
NEAREST, BILINEAR = 0, 1
REDUCE_MEAN, REDUCE_MAX = 0, 1
MATH_F32, MATH_F16X3, MATH_F16 = 0, 1, 2
DT_F32, DT_F16, DT_BF16 = 0, 1, 2


class ConvDesc(Structure):
    """struct arseg_conv_desc (include/arseg_hip.h)."""
    _fields_ = [("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("in_ld", c_int),
                ("Cout", c_int), ("out_ld", c_int), ("res_ld", c_int),
                ("R", c_int), ("S", c_int), ("stride", c_int), ("pad", c_int), ("dil", c_int),
                ("act", c_int), ("prelu_slope", c_float), ("tile_cfg", c_int), ("split_k", c_int),
                ("batch", c_int), ("in_batch_stride", c_int64), ("w_batch_stride", c_int64), ("out_batch_stride", c_int64),
                ("math", c_int), ("upsample2x", c_int), ("range_flag", c_void_p), ("range_limit", c_float)]


_P = c_void_p  # device or host pointer passed as integer
_STREAM = c_void_p

# name -> (restype, argtypes); must list every function of include/arseg_hip.h (tests check this)
PROTOTYPES = {
    "arseg_version": (c_int, []),
    "arseg_status_string": (c_char_p, [c_int]),
    "arseg_local_similar_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_local_weighting_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_local_similar_nhwc_fwd": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_local_weighting_nhwc_fwd": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_warp_fwd": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_mv_resize_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_flow_resize_fwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_warp_mvq_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_creff_fwd": (c_int, [_P] * 8 + [_P, _P, _P, c_int, _P, c_int] + [c_int] * 8 + [_STREAM]),
    "arseg_creff_fwd_ex": (c_int, [_P] * 8 + [_P, _P, _P, c_int, _P, c_int] + [c_int] * 10 + [_STREAM]),
    "arseg_creff_warp_fwd": (c_int, [_P, _P, c_int, c_int, _P] + [_P] * 6 + [_P, c_int, _P, _P, c_int, _P, c_int] + [c_int] * 8 + [_STREAM]),
    "arseg_creff_warp_fwd_ex": (c_int, [_P, _P, c_int, c_int, _P] + [_P] * 6 + [_P, c_int, _P, _P, c_int, _P, c_int] + [c_int] * 11 + [_STREAM]),
    "arseg_to_c8_fwd": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, _STREAM]),
    "arseg_from_c8_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_conv_out_hw": (c_int, [POINTER(ConvDesc), POINTER(c_int), POINTER(c_int)]),
    "arseg_conv2d_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "arseg_conv2d_fwd": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, c_size_t, _STREAM]),
    "arseg_conv2d_find_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "arseg_conv2d_find": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, c_size_t, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_float), _STREAM]),
    "arseg_psp_w2_split_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_float, _STREAM]),
    "arseg_creff_warp_select": (c_int, [c_int] * 12),
    "arseg_peak_stream_copy": (c_int, [_P, _P, c_size_t, _STREAM]),
    "arseg_peak_mfma_f16": (c_int, [_P, c_int, POINTER(c_double), _STREAM]),
    "arseg_gemm_rows16_fwd": (c_int, [_P, _P, _P, c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, c_float, c_int, _STREAM]),
    "arseg_split_rows_fwd": (c_int, [_P, c_int64, _P, c_int64, c_int, c_float, _P, c_float, _STREAM]),
    "arseg_gemm_x3_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, _P, c_int, c_int, c_float, c_int, c_int, _P, c_float, _STREAM]),
    "arseg_gemm_x3_cat_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, _P, _P,
                              c_int, c_float, c_int, c_int, _P, c_float, _STREAM]),
    "arseg_wino43_tiles": (c_int64, [c_int, c_int, c_int, c_int]),
    "arseg_wino43_input_fwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _STREAM]),
    "arseg_wino43_input_split_fwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_float, _STREAM]),
    "arseg_wino43_output_fwd": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _STREAM]),
    "arseg_upconv3x3_tap_gather_fwd": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _STREAM]),
    "arseg_upconv3x3_tap_gather_split_fwd": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_float, _STREAM]),
    "arseg_wino43_pack_weight_host": (c_int, [_P, c_int, c_int, _P]),
    "arseg_packed_k": (c_int, [c_int, c_int, c_int]),
    "arseg_pack_conv_weight_host": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P]),
    "arseg_split_weight_f16x3_host": (c_int, [_P, c_int, c_int, _P, _P]),
    "arseg_fold_bn_host": (c_int, [_P, _P, _P, _P, c_float, _P, c_int, _P, _P]),
    "arseg_pack_dw3x3_host": (c_int, [_P, c_int, _P]),
    "arseg_packed_k16": (c_int, [c_int, c_int, c_int]),
    "arseg_pack_conv_weight16_host": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "arseg_conv2d16_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "arseg_conv2d16_fwd": (c_int, [POINTER(ConvDesc), c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t, _STREAM]),
    "arseg_frame_to_nhwc8_16_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_maxpool3x3s2_16_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_global_mean16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "arseg_global_mean16_fwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _STREAM]),
    "arseg_resize16_fwd": (c_int, [_P, _P] + [c_int] * 11 + [_STREAM]),
    "arseg_scale_add16_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_head16_fwd": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_cast_fwd": (c_int, [_P, c_int, _P, c_int, c_int64, _STREAM]),
    "arseg_warp_mvq16_fwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_warp_mvq16_shared_fwd": (c_int, [_P, c_int64, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_maxpool3x3s2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_adaptive_avgpool_fwd": (c_int, [_P, c_int, _P, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_adaptive_avgpool_blockrow_fwd": (c_int, [_P, c_int, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_psp_pool_matrix_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "arseg_psp_pool_matrix_fwd": (c_int, [_P, c_int, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), _STREAM]),
    "arseg_psp_prior_sum_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), _STREAM]),
    "arseg_global_reduce_fwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_global_reduce_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "arseg_global_reduce_ws_fwd": (c_int, [_P, c_int, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_resize_fwd": (c_int, [_P, _P] + [c_int] * 11 + [_STREAM]),
    "arseg_scale_add_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _STREAM]),
    "arseg_head_fwd": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_frame_to_nhwc4_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_frame_u8_to_nhwc4_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), _STREAM]),
    "arseg_merge_motion_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "arseg_merge_motion_fwd": (c_int, [_P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_nchw_to_nhwc_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _STREAM]),
    "arseg_nhwc_to_nchw_fwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _STREAM]),
    "arseg_argmax_confusion_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_STREAM]),
}

# the SURVEY.md section 8(b) names: aliases with the prototypes of their targets
for _alias, _target in (("arseg_creff_fused_fwd", "arseg_creff_warp_fwd"), ("arseg_conv2d_bn_act_fwd", "arseg_conv2d_fwd"),
                        ("arseg_pack_weights", "arseg_pack_conv_weight_host"), ("arseg_maxpool3x3s2", "arseg_maxpool3x3s2_fwd"),
                        ("arseg_adaptive_avgpool", "arseg_adaptive_avgpool_fwd"), ("arseg_global_reduce", "arseg_global_reduce_fwd"),
                        ("arseg_resize", "arseg_resize_fwd"), ("arseg_scale_add", "arseg_scale_add_fwd")):
    PROTOTYPES[_alias] = PROTOTYPES[_target]

_lib = None


class ArsegError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises if it has not been built -- see __graft_entry__.build()."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ArsegError(f"{LIB_PATH} is missing: build it with `make -C ar-seg_amd/csrc` "
                         f"(or `python -c 'import __graft_entry__ as g; g.build()'`).  There is no fallback path.")
    # PyTorch first: it ships its own libamdhip64.so.7 (+ HSA runtime), and the first copy of that SONAME a process loads is the one every
    # later library binds to.  Loading this library before torch pulls in /opt/rocm's runtime instead; torch then runs on a runtime stack
    # it was not built with and every launch fails with hipErrorNoDevice (seen with build() and smoke() in one process).
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this binding drifted apart
        fn.restype, fn.argtypes = res, args
    if lib.arseg_version() != ABI_VERSION:
        raise ArsegError(f"ABI version mismatch: library reports {lib.arseg_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status == ARSEG_OK:
        return
    msg = load().arseg_status_string(int(status)).decode()
    raise ArsegError(f"{what} failed: status {status} ({msg})")
