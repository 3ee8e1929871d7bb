"""Tensor-level wrappers over the C ABI (include/arseg_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation is a call into
libarseg_hip.so.  All functions raise if a tensor is not a float32 (or, on the 16-bit storage path, fp16 / bf16) CUDA(HIP) tensor --
there is no CPU or PyTorch fallback.

Internal activation layout is NHWC: tensors of shape ``[N, H, W, C]`` (C contiguous).  The helpers
``to_nhwc`` / ``to_nchw`` convert at the API boundary (the reference's interface is NCHW).

The package is split by concern:
  _config   the knobs (``config``, ``configure``), one validation for environment and run-time changes
  _profile  per-launch HIP-event timing (``profile``) and the launch funnel every wrapper goes through
  _plans    the per-shape plan cache + its JSON mirror, the timing helpers of the tuner
  _base     pointer / stream marshalling, argument checks, workspace, the operand-range word
  layers    layout helpers, localAttention pair, ingest, small layers, evaluator tail
  creff     warp + CReFF
  conv      the conv engine (route selection, split rows, Winograd, tap decomposition, 16-bit path, PSP bottleneck)
"""
from . import _base, _config, _plans, _profile
from ._base import _DT16, _need_gpu, _need_gpu16, _nhwc_ld, _ptr, _range_word, _stream, is16, range_tripped, workspace
from ._config import Config, config, configure, set_conv_math
from ._plans import _conv_plans, _time
from ._profile import profile
from .conv import (SplitRows, _conv1x1_x3, _conv2d16, _conv_up2_taps, _conv_wino, conv2d, gemm_rows16, gemm_x3_enabled, igemm3_enabled,
                   psp_bottleneck_x3, psp_x3_foldable, split_rows)
from .creff import creff, creff_warp, creff_warp_kernel, flow_resize, mv_resize, warp, warp_mvq
from .layers import (adaptive_avgpool, argmax_confusion, as_nchw, cast, frame_ingest, frame_to_nhwc4, frame_u8_to_nhwc4, from_c8, global_reduce, head,
                     is_nhwc_view, local_similar, local_weighting, maxpool3x3s2, merge_motion, psp_pool_matrix, psp_prior_sum, resize_nchw, resize_nhwc,
                     scale_add, to_c8, to_nchw_contiguous, to_nhwc)

_LEGACY_SWITCHES = {"_AUTOTUNE": "AUTOTUNE", "_math": "math", "_RANGE_MODE": "RANGE_MODE", "_RANGE_GUARD": "RANGE_GUARD", "_NATIVE_FIND": "NATIVE_FIND",
                    "_WINOGRAD": "WINOGRAD", "_UP2_TAPS": "UP2_TAPS", "_GEMM_X3": "GEMM_X3", "_IGEMM3": "IGEMM3", "_PLAN_FILE": "PLAN_FILE"}


def __getattr__(name):
    """Read-only views of the switches under their former module-level names (tests, bench.py)."""
    if name in _LEGACY_SWITCHES:
        return getattr(_config.sw, _LEGACY_SWITCHES[name])
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
