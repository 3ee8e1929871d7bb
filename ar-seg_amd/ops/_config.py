"""Configuration of the Python host layer: every knob in one object (the library reads no environment variables).

``config`` is initialised once from the environment, changed at run time with ``configure(...)``; ``sw`` holds the switches the hot
paths read (derived from ``config`` by ``_apply_config``).  Both ways in validate with the same rules: a typo in an A/B run raises
instead of silently measuring the default kernel."""
from __future__ import annotations

import dataclasses
import os
from typing import Optional

from .. import _lib

# Which MFMA back end evaluates the fp32 GEMMs (include/arseg_hip.h: enum arseg_math): "f16x3" = fp32 emulated with three
# fp16 MFMAs on hi/lo-split operands (22-bit significands, fp32 accumulate), "f32" = the fp32 MFMA.
MATH_NAMES = {"f32": _lib.MATH_F32, "f16x3": _lib.MATH_F16X3, "f16": _lib.MATH_F16}      # "f16": reduced precision (plain fp16 operands)
_CHOICES = {
    "conv_math": tuple(MATH_NAMES),
    "conv_find": ("native", "python"),
    "conv_range_guard": ("device", "host", "off"),
    "conv_gemm_x3": (True, False, "wino"),
    "conv_igemm3": (True, False),
    "creff_impl": ("", "mfma", "valu", "fused"),
    "creff_tile_rows": (0, 8, 16),
    "creff_warp_impl": ("", "roll", "tiles"),
}
_NONNEG_INT = ("creff_seg_rows", "creff_max_wgs", "lr_subbatch")


@dataclasses.dataclass
class Config:
    """``ops.config`` -- initialised once from the environment (variable in brackets), changed at run time with ``ops.configure(...)``.

    conv_math        [ARSEG_CONV_MATH = f16x3 | f32 | f16]     MFMA back end of the fp32 conv GEMMs (f16x3: hi/lo split, 3 fp16 MFMAs per
                     product; f32: v_mfma_f32_32x32x2_f32; f16: reduced-precision comparison point)
    conv_autotune    [ARSEG_CONV_AUTOTUNE = 1 | 0]              per-shape plans timed on first use (0: the library's tile heuristic)
    conv_find        [ARSEG_CONV_FIND = native | python]        who times the candidate plans: arseg_conv2d_find or the host loop
    conv_winograd    [ARSEG_CONV_WINOGRAD = 1 | 0]              let the tuner consider Winograd F(4x4,3x3)
    conv_wino_margin [ARSEG_CONV_WINO_MARGIN = f]              Winograd is taken when f x its time (three launches, 6x the HBM bytes of the direct
                                                               conv) is below the best direct plan's, both timed alone: in a step that shares HBM
                                                               with other lanes the transforms run slower than alone
    conv_up2_taps    [ARSEG_CONV_UP2_TAPS = 1 | 0]              let the tuner consider the tap decomposition for convs after a x2 upsample
    conv_gemm_x3     [ARSEG_CONV_GEMM_X3 = 1 | wino | 0]        let the tuner consider the LDS-DMA GEMM on pre-split operands (csrc/gemm_x3.hip): for
                     the Winograd GEMMs, the 1x1 convs and the PSP bottleneck -> up_1 chain on split rows (1), the Winograd GEMMs only (wino)
    conv_igemm3      [ARSEG_CONV_IGEMM3 = 1 | 0]                let 3x3 stride-1 convs run as an implicit GEMM of the LDS-DMA kernel on zero-bordered
                     ("padded") activations: one K step per (tap, 32-channel group), the tap is a row offset of the DMA source
    conv_range_guard [ARSEG_CONV_RANGE_GUARD = device | host | 0]   operand range of the f16x3 back end: sticky device word read by
                     ops.range_tripped() (default) / amax + host sync per conv with an immediate fp32 fallback / off
    conv_plan_file   [ARSEG_CONV_PLAN_FILE = <json>]            persist the tuned plans
    creff_impl       [ARSEG_CREFF_IMPL = mfma | valu | fused]   pin one CReFF kernel for C >= 128 (A/B measurements, tests)
    creff_tile_rows  [ARSEG_CREFF_TY = 8 | 16]                  pin the tile height of the matrix-core CReFF kernel
    creff_warp_impl  [ARSEG_CREFF_WARP_IMPL = roll | tiles]     fused warp + CReFF kernel for C = 64: the rolling kernel (csrc/creff_roll.hip, default) or
                     the 16 x 16 tile kernel of rounds 2-3 (csrc/creff_rr.hip)
    creff_seg_rows   [ARSEG_CREFF_SEG_ROWS = n]                 fixed strip segments of n rows for the rolling kernel (0: its balanced default schedule)
    creff_max_wgs    [ARSEG_CREFF_MAX_WGS = n]                  upper bound on the rolling kernel's persistent workgroups (0: one per compute unit)
    lr_subbatch      [ARSEG_LR_SUBBATCH = n]                    evaluate the LR batch of a GOP in slices of n frames (bounds the working set)
    aux_outputs      [ARSEG_AUX_OUTPUTS = 0 | 1]                the fast paths (evaluation.alter_res_*) also evaluate the training-only auxiliary outputs that
                     forward_phase1 returns and evaluation.py:190-191 discards (default 0: skipped; forward() / forward_phase1() always return them)
    (ARSEG_HIP_LIB = <path> selects an alternative library build; it is read by _lib before anything is loaded.)"""
    conv_math: str = "f16x3"
    conv_autotune: bool = True
    conv_find: str = "native"
    conv_winograd: bool = True
    conv_wino_margin: float = 1.0
    conv_up2_taps: bool = True
    conv_gemm_x3: object = True          # True | "wino" | False
    conv_igemm3: bool = True
    conv_range_guard: str = "device"
    conv_plan_file: Optional[str] = None
    creff_impl: str = ""
    creff_tile_rows: int = 0
    creff_warp_impl: str = ""
    creff_seg_rows: int = 0
    creff_max_wgs: int = 0
    lr_subbatch: int = 0
    aux_outputs: bool = False

    @classmethod
    def from_env(cls):
        e = os.environ.get

        def num(name, conv, default):
            raw = e(name, "")
            if raw == "":
                return default
            try:
                return conv(raw)
            except ValueError:
                raise _lib.ArsegError(f"{name}={raw!r} is not a number") from None

        c = cls(conv_math=e("ARSEG_CONV_MATH", "f16x3"), conv_autotune=e("ARSEG_CONV_AUTOTUNE", "1") != "0",
                conv_find=e("ARSEG_CONV_FIND", "native"), conv_winograd=e("ARSEG_CONV_WINOGRAD", "1") != "0",
                conv_wino_margin=num("ARSEG_CONV_WINO_MARGIN", float, 1.0),
                conv_up2_taps=e("ARSEG_CONV_UP2_TAPS", "1") != "0", conv_gemm_x3={"0": False, "wino": "wino"}.get(e("ARSEG_CONV_GEMM_X3", "1"), True),
                conv_igemm3=e("ARSEG_CONV_IGEMM3", "1") != "0",
                conv_range_guard={"1": "host", "0": "off"}.get(e("ARSEG_CONV_RANGE_GUARD", "device"), e("ARSEG_CONV_RANGE_GUARD", "device")),
                conv_plan_file=e("ARSEG_CONV_PLAN_FILE"), creff_impl=e("ARSEG_CREFF_IMPL", ""), creff_tile_rows=num("ARSEG_CREFF_TY", int, 0),
                creff_warp_impl=e("ARSEG_CREFF_WARP_IMPL", ""), creff_seg_rows=num("ARSEG_CREFF_SEG_ROWS", int, 0),
                creff_max_wgs=num("ARSEG_CREFF_MAX_WGS", int, 0), lr_subbatch=num("ARSEG_LR_SUBBATCH", int, 0),
                aux_outputs=e("ARSEG_AUX_OUTPUTS", "0") not in ("", "0"))
        validate(dataclasses.asdict(c), source="environment")
        return c


def validate(kw, source="configure"):
    """The one set of rules for both ways in (the environment used to accept what configure() rejects)."""
    names = {f.name for f in dataclasses.fields(Config)}
    for k, v in kw.items():
        if k not in names:
            raise _lib.ArsegError(f"unknown configuration key {k!r}")
        if k in _CHOICES and v not in _CHOICES[k]:
            raise _lib.ArsegError(f"{k} must be one of {list(_CHOICES[k])}, got {v!r} ({source})")
        if k == "conv_wino_margin" and not (isinstance(v, (int, float)) and not isinstance(v, bool) and v > 0):
            raise _lib.ArsegError(f"conv_wino_margin must be a positive number, got {v!r} ({source})")
        if k in _NONNEG_INT and not (isinstance(v, int) and not isinstance(v, bool) and v >= 0):
            raise _lib.ArsegError(f"{k} must be a non-negative integer, got {v!r} ({source})")


class _Switches:
    """What the hot paths read (attribute lookups at call time, so configure() takes effect at once)."""
    __slots__ = ("AUTOTUNE", "math", "RANGE_MODE", "RANGE_GUARD", "NATIVE_FIND", "WINOGRAD", "UP2_TAPS", "GEMM_X3", "IGEMM3", "PLAN_FILE")


config = Config.from_env()
sw = _Switches()
sw.PLAN_FILE = config.conv_plan_file
_plan_file_listeners = []          # _plans registers its cache here: a new plan file is merged in when the knob changes


def _apply_config():
    """Push ``config`` into the switches."""
    if config.conv_plan_file != sw.PLAN_FILE:          # a new plan file: its plans join the cache, later plans are mirrored to it
        sw.PLAN_FILE = config.conv_plan_file
        for fn in _plan_file_listeners:
            fn()
    sw.AUTOTUNE, sw.math = config.conv_autotune, MATH_NAMES[config.conv_math]
    sw.RANGE_MODE = config.conv_range_guard
    sw.RANGE_GUARD = sw.RANGE_MODE == "host"
    sw.NATIVE_FIND, sw.WINOGRAD, sw.UP2_TAPS = config.conv_find != "python", config.conv_winograd, config.conv_up2_taps
    sw.GEMM_X3, sw.IGEMM3 = config.conv_gemm_x3, config.conv_igemm3


_apply_config()


def configure(**kw):
    """Change knobs of ``ops.config`` at run time (names as in Config); returns the previous values of the ones changed."""
    validate(kw)                     # everything is checked before anything changes
    prev = {k: getattr(config, k) for k in kw}
    for k, v in kw.items():
        setattr(config, k, v)
    try:
        _apply_config()
    except Exception:
        for k, v in prev.items():
            setattr(config, k, v)
        _apply_config()
        raise
    return prev


def set_conv_math(name: str) -> str:
    """Select the conv arithmetic back end ("f32" | "f16x3" | "f16") for subsequent launches; returns the previous one."""
    prev = [k for k, v in MATH_NAMES.items() if v == sw.math][0]
    sw.math = MATH_NAMES[name]
    config.conv_math = name
    return prev
