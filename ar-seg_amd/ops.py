"""Tensor-level wrappers over the C ABI (include/arseg_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation is a call into
libarseg_hip.so.  All functions raise if a tensor is not a float32 CUDA(HIP) tensor -- there is no
CPU or PyTorch fallback.

Internal activation layout is NHWC: tensors of shape ``[N, H, W, C]`` (C contiguous).  The helpers
``to_nhwc`` / ``to_nchw`` convert at the API boundary (the reference's interface is NCHW).
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, check


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------------------------
# configuration: every knob of the Python host layer in one object (the library reads no environment variables)
# ----------------------------------------------------------------------------------------------
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
    conv_range_guard [ARSEG_CONV_RANGE_GUARD = device | host | 0]   operand range of the f16x3 back end: sticky device word read by
                     ops.range_tripped() (default) / amax + host sync per conv with an immediate fp32 fallback / off
    conv_plan_file   [ARSEG_CONV_PLAN_FILE = <json>]            persist the tuned plans
    creff_impl       [ARSEG_CREFF_IMPL = mfma | valu]           pin one of the two CReFF kernels for C >= 128 (A/B measurements, tests)
    creff_tile_rows  [ARSEG_CREFF_TY = 8 | 16]                  pin the tile height of the matrix-core CReFF kernel
    creff_warp_impl  [ARSEG_CREFF_WARP_IMPL = roll | tiles]     fused warp + CReFF kernel for C = 64: the rolling kernel (csrc/creff_roll.hip, default) or
                     the 16 x 16 tile kernel of rounds 2-3 (csrc/creff_rr.hip)
    creff_seg_rows   [ARSEG_CREFF_SEG_ROWS = n]                 fixed strip segments of n rows for the rolling kernel (0: its balanced default schedule)
    creff_max_wgs    [ARSEG_CREFF_MAX_WGS = n]                  upper bound on the rolling kernel's persistent workgroups (0: one per compute unit)
    lr_subbatch      [ARSEG_LR_SUBBATCH = n]                    evaluate the LR batch of a GOP in slices of n frames (bounds the working set)
    (ARSEG_HIP_LIB = <path> selects an alternative library build; it is read by _lib before anything is loaded.)"""
    conv_math: str = "f16x3"
    conv_autotune: bool = True
    conv_find: str = "native"
    conv_winograd: bool = True
    conv_wino_margin: float = 1.0
    conv_up2_taps: bool = True
    conv_gemm_x3: object = True          # True | "wino" | False
    conv_range_guard: str = "device"
    conv_plan_file: Optional[str] = None
    creff_impl: str = ""
    creff_tile_rows: int = 0
    creff_warp_impl: str = ""
    creff_seg_rows: int = 0
    creff_max_wgs: int = 0
    lr_subbatch: int = 0

    @classmethod
    def from_env(cls):
        e = os.environ.get
        return cls(conv_math=e("ARSEG_CONV_MATH", "f16x3"), conv_autotune=e("ARSEG_CONV_AUTOTUNE", "1") != "0",
                   conv_find=e("ARSEG_CONV_FIND", "native"), conv_winograd=e("ARSEG_CONV_WINOGRAD", "1") != "0", conv_wino_margin=float(e("ARSEG_CONV_WINO_MARGIN", "1.0") or 1.0),
                   conv_up2_taps=e("ARSEG_CONV_UP2_TAPS", "1") != "0", conv_gemm_x3={"0": False, "wino": "wino"}.get(e("ARSEG_CONV_GEMM_X3", "1"), True),
                   conv_range_guard={"1": "host", "host": "host", "0": "off", "off": "off"}.get(e("ARSEG_CONV_RANGE_GUARD", "device"), "device"),
                   conv_plan_file=e("ARSEG_CONV_PLAN_FILE"), creff_impl=e("ARSEG_CREFF_IMPL", ""), creff_tile_rows=int(e("ARSEG_CREFF_TY", "0") or 0),
                   creff_warp_impl=e("ARSEG_CREFF_WARP_IMPL", ""), creff_seg_rows=int(e("ARSEG_CREFF_SEG_ROWS", "0") or 0), creff_max_wgs=int(e("ARSEG_CREFF_MAX_WGS", "0") or 0),
                   lr_subbatch=int(e("ARSEG_LR_SUBBATCH", "0") or 0))


config = Config.from_env()


def configure(**kw):
    """Change knobs of ``ops.config`` at run time (names as in Config); returns the previous values of the ones changed."""
    for k, v in kw.items():          # validate everything before anything changes
        if not hasattr(config, k):
            raise _lib.ArsegError(f"unknown configuration key {k!r}")
        if k == "conv_wino_margin" and not (isinstance(v, (int, float)) and v > 0):
            raise _lib.ArsegError(f"conv_wino_margin must be a positive number, got {v!r}")
        if k == "conv_math" and v not in _MATH_NAMES:
            raise _lib.ArsegError(f"conv_math must be one of {sorted(_MATH_NAMES)}, got {v!r}")
        if k == "creff_warp_impl" and v not in ("", "roll", "tiles"):
            raise _lib.ArsegError(f"creff_warp_impl must be '', 'roll' or 'tiles', got {v!r}")
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


# ----------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py): HIP events on the launch stream around every ABI call
# ----------------------------------------------------------------------------------------------
_profile = None


class profile:
    """``with ops.profile() as prof: ...`` records (op name, algorithmic flops, algorithmic bytes, ms) per launch.
    Events are recorded on torch's current stream, which is the stream handed to the library.  ``only`` = a set of op names: time
    just those (the others launch without events, so that concurrent streams keep the GPU as busy as in an un-instrumented run)."""

    def __init__(self, only=None):
        self.only = None if only is None else frozenset(only)

    def __enter__(self):
        global _profile
        self.records = []
        _profile = self
        return self

    def __exit__(self, *exc):
        global _profile
        _profile = None
        torch.cuda.synchronize()
        self.rows = [(name, flops, nbytes, s.elapsed_time(e)) for name, flops, nbytes, s, e, _ in self.records]
        self.tags = [tag for *_, tag in self.records]
        return False

    def summary(self):
        out = {}
        for name, flops, nbytes, ms in self.rows:
            r = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0, "bytes": 0})
            r["launches"] += 1
            r["ms"] += ms
            r["flops"] += flops
            r["bytes"] += nbytes
        return out

    def layers(self):
        """Per conv layer (one row per distinct shape + plan): every launch made on behalf of the layer (GEMM / patch kernel, Winograd
        transforms, tap gather, materialised upsample), with the reference's direct-conv FLOP count of the layer."""
        out = {}
        for (name, flops, nbytes, ms), tag in zip(self.rows, self.tags):
            if tag is None:
                continue
            r = out.setdefault(tag, {"calls": 0, "ms": 0.0, "kernels": {}})
            r["ms"] += ms
            r["kernels"][name] = r["kernels"].get(name, 0.0) + ms
            r["calls"] += name == "conv2d"
        rows = []
        for (N, H, W, cin, cout, k, stride, dil, up2, plan, flops), r in out.items():
            calls = max(r["calls"], 1)
            rows.append({"N": N, "H": H, "W": W, "cin": cin, "cout": cout, "k": k, "stride": stride, "dil": dil, "up2": up2, "plan": plan,
                         "calls": calls, "us_per_call": 1e3 * r["ms"] / calls, "gflop_per_call": flops / 1e9,
                         "tflops": flops * calls / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else None,
                         "us_by_kernel": {kk: 1e3 * v / calls for kk, v in r["kernels"].items()}})
        return rows


_layer_tag = None      # set by conv2d while it launches on behalf of one layer (profile.layers())


def _launch(name, fn, *args, flops=0, nbytes=0):
    if _profile is None or (_profile.only is not None and name not in _profile.only):
        check(fn(*args), name)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    check(fn(*args), name)
    e.record()
    _profile.records.append((name, flops, nbytes, s, e, _layer_tag))


def _need_gpu(*tensors, dtype=torch.float32):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.ArsegError("arseg_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if dtype is not None and t.dtype != dtype:
            raise _lib.ArsegError(f"expected {dtype}, got {t.dtype}")


_DT16 = {torch.float16: _lib.DT_F16, torch.bfloat16: _lib.DT_BF16}


def is16(t: torch.Tensor) -> bool:
    """True for the 16-bit storage path (BASELINE configs[2] / configs[4]): fp16 or bf16 NHWC tensors."""
    return t.dtype in _DT16


def _need_gpu16(*tensors):
    dt = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.ArsegError("arseg_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if t.dtype not in _DT16 or (dt is not None and t.dtype != dt):
            raise _lib.ArsegError(f"16-bit path: expected tensors of one 16-bit dtype, got {t.dtype}")
        dt = t.dtype
    return _DT16[dt]


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 <-> fp16 / bf16 element conversion on the GPU (round to nearest even), any shape with numel % 8 == 0."""
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    code = lambda d: _lib.DT_F32 if d == torch.float32 else _DT16[d]
    _launch("cast", _lib.load().arseg_cast_fwd, _ptr(x), code(x.dtype), _ptr(out), code(dtype), x.numel(), _stream())
    return out


def frame_ingest(img: torch.Tensor, h: int, w: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """NCHW RGB frame -> the conv engine's input at (h,w): NHWC4 fp32, or NHWC8 fp16 / bf16 on the 16-bit storage path."""
    if dtype == torch.float32:
        return frame_to_nhwc4(img, h, w)
    _need_gpu(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    if C != 3:
        raise _lib.ArsegError("frame_ingest expects 3 input channels")
    out = torch.empty((N, h, w, 8), dtype=dtype, device=img.device)
    _launch("frame_to_nhwc8", _lib.load().arseg_frame_to_nhwc8_16_fwd, _ptr(img), _ptr(out), _DT16[dtype], N, H, W, h, w, _stream())
    return out


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# ----------------------------------------------------------------------------------------------
# workspace (caller-owned, as the ABI requires): one growing buffer per device
# ----------------------------------------------------------------------------------------------
_workspaces = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    # one buffer per (device, stream): split-K partials of convs running concurrently on different streams must not alias
    if torch.cuda.is_current_stream_capturing():
        # Inside a HIP-graph capture the buffer comes from that graph's private pool and its address is baked into the graph: it belongs
        # to the graph alone (the graph's pool keeps it alive), never to this cache -- a second graph captured on the same (singleton)
        # capture stream, or eager code on a recycled stream id, would otherwise share split-K partial sums with it.
        return torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    key = (torch.device(device).index or 0, torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# ----------------------------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------------------------
def is_nhwc_view(x: torch.Tensor) -> bool:
    """True if logical-NCHW ``x`` is physically NHWC-contiguous (channels_last or a permuted NHWC tensor)."""
    N, C, H, W = x.shape
    return x.stride() == (H * W * C, 1, W * C, C)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Logical NCHW tensor -> physical NHWC tensor [N,H,W,C] (zero-copy when already channels_last)."""
    if is16(x):
        _need_gpu16(x)
        return x.permute(0, 2, 3, 1) if is_nhwc_view(x) else x.permute(0, 2, 3, 1).contiguous()
    _need_gpu(x)
    N, C, H, W = x.shape
    if is_nhwc_view(x):
        return x.permute(0, 2, 3, 1)
    x = x.contiguous()
    out = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    _launch("nchw_to_nhwc", _lib.load().arseg_nchw_to_nhwc_fwd, _ptr(x), _ptr(out), N, C, H * W, C, _stream())
    return out


def as_nchw(x_nhwc: torch.Tensor) -> torch.Tensor:
    """Physical NHWC [N,H,W,C] -> logical NCHW view (channels_last strides, no copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


def to_nchw_contiguous(x_nhwc: torch.Tensor) -> torch.Tensor:
    _need_gpu(x_nhwc)
    N, H, W, C = x_nhwc.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x_nhwc.device)
    _launch("nhwc_to_nchw", _lib.load().arseg_nhwc_to_nchw_fwd, _ptr(x_nhwc), C, _ptr(out), N, C, H * W, _stream())
    return out


def to_c8(x: torch.Tensor, layout: int) -> torch.Tensor:
    """NCHW-contiguous [N,C,H,W] or NHWC [N,H,W,C] -> channel-blocked [N,C/8,H,W,8]."""
    _need_gpu(x)
    if layout == _lib.NCHW:
        N, C, H, W = x.shape
        x = x.contiguous()
        ld = 0
    else:
        N, H, W, C = x.shape
        x = x.contiguous()
        ld = C
    out = torch.empty((N, C // 8, H, W, 8), dtype=torch.float32, device=x.device)
    _launch("to_c8", _lib.load().arseg_to_c8_fwd, _ptr(x), layout, ld, _ptr(out), N, C, H * W, _stream())
    return out


def from_c8(x: torch.Tensor, layout: int) -> torch.Tensor:
    _need_gpu(x)
    N, CB, H, W, _ = x.shape
    C = CB * 8
    shape = (N, C, H, W) if layout == _lib.NCHW else (N, H, W, C)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    _launch("from_c8", _lib.load().arseg_from_c8_fwd, _ptr(x), _ptr(out), layout, C, N, C, H * W, _stream())
    return out


# ----------------------------------------------------------------------------------------------
# localAttention pair
# ----------------------------------------------------------------------------------------------
def _is_cl(t: torch.Tensor) -> bool:
    return t.dim() == 4 and t.shape[1] > 1 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


def local_similar(q: torch.Tensor, k: torch.Tensor, kH: int, kW: int) -> torch.Tensor:
    """localAttention.similar_forward; channels_last inputs go to the NHWC variant without a layout change."""
    _need_gpu(q, k)
    N, C, H, W = q.shape
    out = torch.empty((N, H, W, kH * kW), dtype=torch.float32, device=q.device)
    if _is_cl(q) and _is_cl(k):
        _launch("local_similar", _lib.load().arseg_local_similar_nhwc_fwd, _ptr(q), _ptr(k), C, _ptr(out), N, C, H, W, kH, kW, _stream())
        return out
    q, k = q.contiguous(), k.contiguous()
    _launch("local_similar", _lib.load().arseg_local_similar_fwd, _ptr(q), _ptr(k), _ptr(out), N, C, H, W, kH, kW, _stream())
    return out


def local_weighting(v: torch.Tensor, w: torch.Tensor, kH: int, kW: int) -> torch.Tensor:
    """localAttention.weighting_forward; a channels_last ``v`` gives a channels_last result through the NHWC variant."""
    _need_gpu(v, w)
    w = w.contiguous()
    N, C, H, W = v.shape
    if _is_cl(v):
        out = torch.empty_like(v, memory_format=torch.channels_last)
        _launch("local_weighting", _lib.load().arseg_local_weighting_nhwc_fwd, _ptr(v), _ptr(w), C, _ptr(out), N, C, H, W, kH, kW, _stream())
        return out
    v = v.contiguous()
    out = torch.empty_like(v)
    _launch("local_weighting", _lib.load().arseg_local_weighting_fwd, _ptr(v), _ptr(w), _ptr(out), N, C, H, W, kH, kW, _stream())
    return out


# ----------------------------------------------------------------------------------------------
# warp / motion vectors
# ----------------------------------------------------------------------------------------------
def warp(feature: torch.Tensor, flow: torch.Tensor, layout: int, out_layout: Optional[int] = None) -> torch.Tensor:
    """feature NCHW-contiguous [N,C,H,W] (layout=NCHW) or NHWC [N,H,W,C]; flow [N,H,W,2] f32/f64."""
    _need_gpu(feature)
    _need_gpu(flow, dtype=None)
    if flow.dtype not in (torch.float32, torch.float64):
        raise _lib.ArsegError(f"flow must be float32 or float64, got {flow.dtype}")
    feature, flow = feature.contiguous(), flow.contiguous()
    if layout == _lib.NCHW:
        N, C, H, W = feature.shape
    else:
        N, H, W, C = feature.shape
    out_layout = layout if out_layout is None else out_layout
    if out_layout == _lib.C8:
        out = torch.empty((N, C // 8, H, W, 8), dtype=torch.float32, device=feature.device)
    else:
        out = torch.empty_like(feature)
    fd = _lib.FLOW_F64 if flow.dtype == torch.float64 else _lib.FLOW_F32
    _launch("warp", _lib.load().arseg_warp_fwd, _ptr(feature), _ptr(flow), fd, _ptr(out), N, C, H, W, layout, out_layout, _stream())
    return out


def mv_resize(mv_q: torch.Tensor, Hp: int, Wp: int) -> torch.Tensor:
    """int16 quarter-pel [N,H,W,2] -> float64 [N,Hp,Wp,2] (evaluation.py:176-180)."""
    _need_gpu(mv_q, dtype=torch.int16)
    mv_q = mv_q.contiguous()
    N, H, W, _ = mv_q.shape
    out = torch.empty((N, Hp, Wp, 2), dtype=torch.float64, device=mv_q.device)
    _launch("mv_resize", _lib.load().arseg_mv_resize_fwd, _ptr(mv_q), _ptr(out), N, H, W, Hp, Wp, _stream())
    return out


def flow_resize(flow: torch.Tensor, Hp: int, Wp: int) -> torch.Tensor:
    """float flow [N,H,W,2] in pixels (fp32 / fp64) -> float64 [N,Hp,Wp,2] (evaluation.py:176-180), any values."""
    _need_gpu(flow, dtype=None)
    if flow.dtype not in (torch.float32, torch.float64):
        raise _lib.ArsegError(f"flow must be float32 or float64, got {flow.dtype}")
    flow = flow.contiguous()
    N, H, W, _ = flow.shape
    out = torch.empty((N, Hp, Wp, 2), dtype=torch.float64, device=flow.device)
    _launch("flow_resize", _lib.load().arseg_flow_resize_fwd, _ptr(flow), _lib.FLOW_F64 if flow.dtype == torch.float64 else _lib.FLOW_F32, _ptr(out),
            N, H, W, Hp, Wp, _stream())
    return out


def warp_mvq(feature_nhwc: torch.Tensor, mv_q: torch.Tensor, out_layout: int = _lib.C8, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MV resize + warp fused: NHWC feature [N,Hp,Wp,C], int16 quarter-pel MVs [N,H,W,2] at frame resolution.
    ``out``: optional contiguous destination (e.g. one frame's slot of a batched C8 buffer)."""
    _need_gpu(feature_nhwc, out)
    _need_gpu(mv_q, dtype=torch.int16)
    feature_nhwc, mv_q = feature_nhwc.contiguous(), mv_q.contiguous()
    N, Hp, Wp, C = feature_nhwc.shape
    _, H, W, _ = mv_q.shape
    shape = (N, C // 8, Hp, Wp, 8) if out_layout == _lib.C8 else (N, Hp, Wp, C)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=feature_nhwc.device)
    elif tuple(out.shape) != shape or not out.is_contiguous():
        raise _lib.ArsegError(f"warp_mvq out must be contiguous with shape {shape}")
    _launch("warp_mvq", _lib.load().arseg_warp_mvq_fwd, _ptr(feature_nhwc), _ptr(mv_q), _ptr(out), N, C, Hp, Wp, H, W, out_layout, _stream(),
            nbytes=N * (2 * 4 * C * Hp * Wp + 4 * H * W))
    return out


# ----------------------------------------------------------------------------------------------
# CReFF
# ----------------------------------------------------------------------------------------------
def creff(hr_c8: torch.Tensor, lr_nhwc: torch.Tensor, attn, head=None, log_softmax: bool = False, kH: int = 7, kW: int = 7
          ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """attn: packing.PackedAttention; head: None or (wf [n_cls,C], bf [n_cls]) device tensors.
    Returns (p in C8 layout, logits NCHW or None)."""
    _need_gpu(hr_c8, lr_nhwc)
    hr_c8, lr_nhwc = hr_c8.contiguous(), lr_nhwc.contiguous()
    N, CB, Hp, Wp, _ = hr_c8.shape
    C = CB * 8
    _, hp, wp, C2 = lr_nhwc.shape
    if C2 != C:
        raise _lib.ArsegError(f"channel mismatch: hr has {C}, lr has {C2}")
    if N > 1 and N * C * Hp * Wp * 4 >= (1 << 31):          # the kernel addresses p_out with 32-bit buffer offsets
        h = N // 2
        a = creff(hr_c8[:h], lr_nhwc[:h], attn, head, log_softmax, kH, kW)
        b = creff(hr_c8[h:], lr_nhwc[h:], attn, head, log_softmax, kH, kW)
        return torch.cat([a[0], b[0]]), (None if a[1] is None else torch.cat([a[1], b[1]]))
    p_out = torch.empty_like(hr_c8)
    logits, wf, bf, n_cls = None, None, None, 0
    if head is not None:
        wf, bf = head
        n_cls = wf.shape[0]
        logits = torch.empty((N, n_cls, Hp, Wp), dtype=torch.float32, device=hr_c8.device)
    # kernel choice: explicit arguments of the ABI; the knobs (A/B measurements, tests) live in ops.config, not in the library
    impl = {"mfma": 1, "valu": 2}.get(config.creff_impl, 0)
    tile_rows = config.creff_tile_rows
    _launch("creff", _lib.load().arseg_creff_fwd_ex, _ptr(hr_c8), _ptr(lr_nhwc), _ptr(attn.wq), _ptr(attn.bq), _ptr(attn.wk), _ptr(attn.bk),
                                      _ptr(attn.wv), _ptr(attn.bv), _ptr(p_out), _ptr(wf), _ptr(bf), n_cls, _ptr(logits),
                                      1 if log_softmax else 0, N, C, Hp, Wp, hp, wp, kH, kW, impl, tile_rows if tile_rows in (8, 16) else 0, _stream(),
            flops=N * Hp * Wp * C * (250 + 2 * n_cls),
            nbytes=4 * N * (2 * C * Hp * Wp + C * hp * wp + n_cls * Hp * Wp))
    return p_out, logits


def creff_warp(refs_nhwc, mv_q: torch.Tensor, lr_nhwc: torch.Tensor, attn, head=None, log_softmax: bool = False, kH: int = 7,
               kW: int = 7, p_layout: int = _lib.C8) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """MV warp + CReFF + head in one kernel (arseg_creff_warp_fwd).

    refs_nhwc: sequence of B un-warped keyframe features, NHWC [Hp,Wp,C] each (frames of one GOP share theirs);
    mv_q: int16 [B,H,W,2]; lr_nhwc: [B,hp,wp,C].  Returns (p in ``p_layout``, logits NCHW or None).  Shapes the fused
    kernel does not cover (C != 64, windows other than 7x7) run as arseg_warp_mvq_fwd + arseg_creff_fwd."""
    if is16(lr_nhwc):
        # 16-bit storage path: the warp reads the 16-bit keyframe feature directly (fp32 C8 out), the small LR feature is converted,
        # the CReFF arithmetic itself stays fp32 (split-fp16 matrix cores): p and the logits come back fp32
        dt = _need_gpu16(lr_nhwc, *refs_nhwc)
        _need_gpu(mv_q, dtype=torch.int16)
        mv_q = mv_q.contiguous()
        B, hp, wp, C = lr_nhwc.shape
        Hp, Wp, _ = refs_nhwc[0].shape
        _, H, W, _ = mv_q.shape
        ref_c8 = torch.empty((B, C // 8, Hp, Wp, 8), dtype=torch.float32, device=lr_nhwc.device)
        refs16 = [r.contiguous() for r in refs_nhwc]
        if B * C * Hp * Wp * 4 < (1 << 31) and all(r.data_ptr() == refs16[0].data_ptr() for r in refs16):
            # the non-keyframes of one GOP share the keyframe feature: one launch for the batch (feature stride 0) instead of one per frame
            _launch("warp_mvq", _lib.load().arseg_warp_mvq16_shared_fwd, _ptr(refs16[0]), 0, dt, _ptr(mv_q), _ptr(ref_c8), B, C, Hp, Wp,
                    H, W, _stream(), nbytes=B * (2 * C * Hp * Wp + 4 * C * Hp * Wp + 4 * H * W))
        else:
            for b in range(B):
                _launch("warp_mvq", _lib.load().arseg_warp_mvq16_fwd, _ptr(refs16[b]), dt, _ptr(mv_q[b:b + 1]), _ptr(ref_c8[b:b + 1]), 1, C, Hp, Wp,
                        H, W, _stream(), nbytes=2 * C * Hp * Wp + 4 * C * Hp * Wp + 4 * H * W)
        p_c8, logits = creff(ref_c8, cast(lr_nhwc, torch.float32), attn, head, log_softmax, kH, kW)
        return (p_c8 if p_layout == _lib.C8 else from_c8(p_c8, _lib.NHWC)), logits
    _need_gpu(lr_nhwc, *refs_nhwc)
    _need_gpu(mv_q, dtype=torch.int16)
    lr_nhwc, mv_q = lr_nhwc.contiguous(), mv_q.contiguous()
    B, hp, wp, C = lr_nhwc.shape
    refs = [r.contiguous() for r in refs_nhwc]
    Hp, Wp, C2 = refs[0].shape
    _, H, W, _ = mv_q.shape
    if len(refs) != B or mv_q.shape[0] != B or C2 != C or any(tuple(r.shape) != (Hp, Wp, C) for r in refs):
        raise _lib.ArsegError("creff_warp: refs / mv_q / lr batch or channel mismatch")
    fused_ok = C == 64 and kH == 7 and kW == 7 and (head is None or head[0].shape[0] <= 32) and C * Hp * Wp * 4 < (1 << 31)
    if not fused_ok:
        ref_c8 = torch.empty((B, C // 8, Hp, Wp, 8), dtype=torch.float32, device=lr_nhwc.device)
        for b in range(B):
            warp_mvq(refs[b].unsqueeze(0), mv_q[b:b + 1], _lib.C8, out=ref_c8[b:b + 1])
        p_c8, logits = creff(ref_c8, lr_nhwc, attn, head, log_softmax, kH, kW)
        return (p_c8 if p_layout == _lib.C8 else from_c8(p_c8, _lib.NHWC)), logits
    per = max(1, min(32, ((1 << 31) - 1) // (C * Hp * Wp * 4)))          # frames per launch: 32-bit buffer offsets, 32 pointers
    if B > per:
        outs = [creff_warp(refs[i:i + per], mv_q[i:i + per], lr_nhwc[i:i + per], attn, head, log_softmax, kH, kW, p_layout)
                for i in range(0, B, per)]
        return torch.cat([o[0] for o in outs]), (None if outs[0][1] is None else torch.cat([o[1] for o in outs]))
    shape = (B, C // 8, Hp, Wp, 8) if p_layout == _lib.C8 else (B, Hp, Wp, C)
    p_out = torch.empty(shape, dtype=torch.float32, device=lr_nhwc.device)
    logits, wf, bf, n_cls = None, None, None, 0
    if head is not None:
        wf, bf = head
        n_cls = wf.shape[0]
        logits = torch.empty((B, n_cls, Hp, Wp), dtype=torch.float32, device=lr_nhwc.device)
    ptrs = (ctypes.c_void_p * B)(*[r.data_ptr() for r in refs])
    impl = {"tiles": 1, "roll": 2}.get(config.creff_warp_impl, 0)
    _launch("creff_warp", _lib.load().arseg_creff_warp_fwd_ex, ptrs, _ptr(mv_q), H, W, _ptr(lr_nhwc), _ptr(attn.wq), _ptr(attn.bq),
            _ptr(attn.wk), _ptr(attn.bk), _ptr(attn.wv), _ptr(attn.bv), _ptr(p_out), p_layout, _ptr(wf), _ptr(bf), n_cls, _ptr(logits),
            1 if log_softmax else 0, B, C, Hp, Wp, hp, wp, kH, kW, impl, max(0, int(config.creff_seg_rows)), max(0, int(config.creff_max_wgs)), _stream(),
            flops=B * Hp * Wp * C * (250 + 2 * n_cls),
            nbytes=B * (4 * (2 * C * Hp * Wp + C * hp * wp + n_cls * Hp * Wp) + 4 * H * W))
    return p_out, logits


# ----------------------------------------------------------------------------------------------
# conv engine
# ----------------------------------------------------------------------------------------------
def _nhwc_ld(t: torch.Tensor) -> int:
    """Channel stride (floats per pixel) of an NHWC tensor that may be a channel slice of a wider NHWC buffer."""
    N, H, W, C = t.shape
    sN, sH, sW, sC = t.stride()
    bad = C > 1 and sC != 1
    if W > 1:
        ld = sW
    elif H > 1:
        ld = sH
    elif N > 1:
        ld = sN
    else:
        ld = max(sW, C)
    bad = bad or ld < C or (H > 1 and sH != W * ld) or (N > 1 and sN != H * W * ld)
    if bad:
        raise _lib.ArsegError(f"tensor is not an NHWC (slice) view: shape {tuple(t.shape)} strides {t.stride()}")
    return ld


# Per-shape launch plans (tile shape, LDS buffering, split-K).  The library's built-in heuristic is good to ~10 %; the
# first time a conv shape is seen on a device the candidates are timed with HIP events and the fastest is cached
# (what MIOpen calls "find").  ARSEG_CONV_AUTOTUNE=0 keeps the heuristic.
_AUTOTUNE = config.conv_autotune
# Which MFMA back end evaluates the fp32 GEMMs (include/arseg_hip.h: enum arseg_math): "f16x3" = fp32 emulated with three
# fp16 MFMAs on hi/lo-split operands (22-bit significands, fp32 accumulate), "f32" = the fp32 MFMA.
_MATH_NAMES = {"f32": _lib.MATH_F32, "f16x3": _lib.MATH_F16X3, "f16": _lib.MATH_F16}      # "f16": reduced precision (plain fp16 operands)
_math = _MATH_NAMES[config.conv_math]


def set_conv_math(name: str) -> str:
    """Select the conv arithmetic back end ("f32" | "f16x3" | "f16") for subsequent launches; returns the previous one."""
    global _math
    prev = [k for k, v in _MATH_NAMES.items() if v == _math][0]
    _math = _MATH_NAMES[name]
    config.conv_math = name
    return prev
# Operand-range safety of the split-fp16 back end (include/arseg_hip.h, ARSEG_MATH_F16X3: the hi/lo pair carries 22 bits up to |x| = 65504
# and clamps beyond 131008; the Winograd route multiplies TRANSFORMED activations, ~10x the input).
#   * default ("device"): every f16x3 conv gets a sticky device word (arseg_conv_desc.range_flag); the kernels set it when an activation
#     they multiply exceeds 65504 -- no host synchronisation, capturable.  The host reads the word once per batch of launches with
#     ops.range_tripped() (one sync) and repeats the batch under ops.set_conv_math("f32") -- evaluation.Eval*Res do, bench.py reports it.
#   * ARSEG_CONV_RANGE_GUARD=host: the round-2 validation mode -- amax of every conv input with one host sync per conv, the layer is
#     evaluated with the fp32 MFMA back end at once (not capturable).   * ARSEG_CONV_RANGE_GUARD=0: off.
_RANGE_MODE = config.conv_range_guard
_RANGE_GUARD = _RANGE_MODE == "host"
_RANGE_LIMIT = 2.0e4
_range_words = {}


def _range_word(device):
    """The device's sticky status word (allocated outside any graph capture; None while capturing before the first eager conv)."""
    idx = torch.device(device).index or 0
    w = _range_words.get(idx)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        w = _range_words[idx] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def _arm_range_watch(d, device):
    if _RANGE_MODE == "device" and d.math == _lib.MATH_F16X3:
        w = _range_word(device)
        if w is not None:
            d.range_flag, d.range_limit = w.data_ptr(), 65504.0


def range_tripped(device=None, reset: bool = True) -> bool:
    """True if an f16x3 conv launched since the last reset multiplied an activation beyond the split-fp16 range (its result may be
    clamped): repeat those launches with set_conv_math("f32").  One device -> host read (synchronises the current stream)."""
    idx = torch.device(device).index or 0 if device is not None else torch.cuda.current_device()
    w = _range_words.get(idx)
    if w is None:
        return False
    hit = bool(int(w.item()) & 1)
    if hit and reset:
        w.zero_()
    return hit


_PLAN_FILE = config.conv_plan_file       # optional: persist tuned plans (skips the trial launches next time)


class _PlanCache(dict):
    """Plans keyed by shape tuples; optionally mirrored to a JSON file (``ops.configure(conv_plan_file=...)`` merges that file in)."""

    def __init__(self):
        super().__init__()
        self.load()

    def load(self):
        if _PLAN_FILE and os.path.exists(_PLAN_FILE):
            import json

            with open(_PLAN_FILE) as f:
                for k, v in json.load(f).items():
                    super().__setitem__(tuple(json.loads(k)), tuple(v) if isinstance(v, list) else v)

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        if _PLAN_FILE:
            import json

            with open(_PLAN_FILE, "w") as f:
                json.dump({json.dumps(list(k)): (list(v) if isinstance(v, tuple) else v) for k, v in self.items()}, f, indent=0)


_conv_plans = _PlanCache()
_NATIVE_FIND = config.conv_find != "python"      # "python": time the candidates from the host loop below instead
_PATCH_CFGS = (13, 14, 15, 16)      # arseg_conv_desc.tile_cfg of the patch-resident 3x3 kernel (the only direct plans with a fused x2 upsample)


def _conv_candidates(ktiles: int, cout: int, m: int, patch_ok: bool = False):
    cands = []
    if patch_ok:                                   # patch-resident 3x3 kernel (13/14: 128-pixel tiles, 15/16: 256; BN 64/128)
        cands += [(15, 1), (13, 1)] + ([(16, 1), (14, 1)] if cout > 64 else [])
    for cfg in (5, 6, 7, 8, 9, 10, 11, 12) + ((17, 18, 19) if _math == _lib.MATH_F16X3 else ()):
        bn = {17: 128, 18: 256, 19: 256}.get(cfg, 128 if cfg in (5, 8, 9, 12) else 64)
        bm = {17: 256, 18: 128, 19: 256}.get(cfg, 128 if cfg in (5, 6, 9, 10) else 64)
        if bn >= 128 and cfg >= 17 and cout < bn:
            continue
        if bn == 128 and cout <= 64:
            continue
        if bm == 128 and m <= 64:
            continue
        for sk in (1, 2, 3, 4, 6, 8):
            if sk > 1 and (ktiles // sk < 4 or cout % 4):
                continue
            cands.append((cfg, sk))
    return cands


class SplitRows:
    """An NHWC activation tensor stored as split rows -- per 32 channels 32 hi fp16 halves then 32 lo halves, the operand format of
    arseg_gemm_x3_fwd (include/arseg_hip.h) -- in a float32-typed buffer ``t`` of the logical shape (the same 4 bytes per value).
    Produced by ``conv2d(..., out_split=True)`` / ``split_rows``; consumed by ``conv2d`` (1x1 convs and the tap-decomposed conv after a
    x2 upsample).  ``float()`` gives the fp32 tensor back (torch ops on the device; only fallback paths need it)."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t

    shape = property(lambda self: self.t.shape)
    device = property(lambda self: self.t.device)

    def float(self):
        n, h, w, c = self.t.shape
        hl = self.t.view(torch.float16).view(n, h, w, c // 32, 2, 32).float()
        return (hl[..., 0, :] + hl[..., 1, :]).reshape(n, h, w, c)


def gemm_x3_enabled() -> bool:
    return _GEMM_X3 is True and _math == _lib.MATH_F16X3 and not _RANGE_GUARD


def split_rows(x: torch.Tensor) -> SplitRows:
    """fp32 NHWC [N,H,W,C] (C % 32 == 0; may be a channel slice) -> SplitRows: one memory-bound pass (arseg_split_rows_fwd) that also
    carries the operand range watch of the GEMM that will consume it."""
    _need_gpu(x)
    n, h, w, c = x.shape
    t = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    rw = _range_word(x.device) if _RANGE_MODE == "device" else None
    _launch("split_rows", _lib.load().arseg_split_rows_fwd, _ptr(x), _nhwc_ld(x), _ptr(t), n * h * w, c, 1.0, _ptr(rw), 65504.0, _stream())
    return SplitRows(t)


def _x3_eligible(pc, cin, residual, up2) -> bool:
    return (gemm_x3_enabled() and not up2 and pc.R == 1 and pc.S == 1 and pc.stride == 1 and pc.pad == 0 and cin % 32 == 0 and cin == pc.cin_pad
            and pc.cout % 4 == 0)


def _conv1x1_x3(x, pc, residual=None, out=None, out_split=False, cfg=None, record=True):
    """1x1 stride-1 conv on the LDS-DMA GEMM (csrc/gemm_x3.hip).  x: SplitRows, or fp32 NHWC (split by a pre-pass first).  cfg None: the
    tile shape is timed on first use per (M, K, N)."""
    lib = _lib.load()
    xs = x if isinstance(x, SplitRows) else split_rows(x)
    N, H, W, Cin = xs.shape
    M, Cout, dev = N * H * W, pc.cout, xs.device
    if out_split and Cout % 32:
        out_split = False                    # split rows come in groups of 32 channels: such a layer writes plain fp32
    if out_split:
        out = torch.empty((N, H, W, Cout), dtype=torch.float32, device=dev)
    elif out is None:
        out = torch.empty((N, H, W, Cout), dtype=torch.float32, device=dev)
    rw = _range_word(dev) if (out_split and _RANGE_MODE == "device") else None

    def run(c, rec):
        args = (_ptr(xs.t), _ptr(pc.w_h3), _ptr(out), M, Cout, Cin, Cout if out_split else _nhwc_ld(out), 1, 0, 0, 0, _ptr(pc.scale_h3), _ptr(pc.bias),
                _ptr(residual), _nhwc_ld(residual) if residual is not None else 0, pc.act, pc.slope, 1 if out_split else 0, c, _ptr(rw), 65504.0, _stream())
        if rec:
            _launch("conv2d", lib.arseg_gemm_x3_fwd, *args, flops=2 * M * Cin * Cout)
        else:
            check(lib.arseg_gemm_x3_fwd(*args), "gemm_x3")

    if cfg is None:
        key = ("x3", dev.index, M, Cin, Cout, bool(out_split), residual is not None)
        cfg = _conv_plans.get(key)
        if cfg is None:
            if not _AUTOTUNE or torch.cuda.is_current_stream_capturing():
                cfg = 0                      # no timing loop inside a graph capture / with the tuner off: the 128 x 128 tile (not cached)
            else:
                best_t = float("inf")
                for c in range(7):
                    try:
                        t = _time(lambda: run(c, False))
                    except _lib.ArsegError:  # a tile shape this problem does not admit
                        continue
                    if t < best_t:
                        cfg, best_t = c, t
                if cfg is None:
                    raise _lib.ArsegError(f"gemm_x3: no tile configuration accepts M={M} K={Cin} N={Cout}")
                _conv_plans[key] = cfg
    run(cfg, record)
    return SplitRows(out) if out_split else out


def conv2d(x: torch.Tensor, pc, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           tile_cfg: int = 0, split_k: int = 0, up2: bool = False, out_split: bool = False):
    """x NHWC [N,H,W,Cin_pad] (may be a channel slice); pc: packing.PackedConv; out: optional NHWC (slice) view.
    tile_cfg / split_k: 0 = use the cached per-shape plan (autotuned on first use).
    up2: the conv input is the x2 bilinear (align_corners=False) upsample of ``x`` (PSPUpsample, model/pspnet.py:43-46);
    the Winograd route applies it inside its input transform, the patch-resident direct plans while they stage their input patch;
    the GEMM-tile plans materialise it first."""
    global _layer_tag
    if isinstance(x, SplitRows):
        # an activation the producer already wrote as split rows: 1x1 convs go straight to the LDS-DMA GEMM, a 3x3 conv after a x2 upsample
        # to its tap decomposition (whose low-resolution 1x1 conv is such a GEMM); anything else reads the fp32 form
        n_, h_, w_, c_ = x.shape
        if not up2 and _x3_eligible(pc, c_, residual, False):
            return _conv1x1_x3(x, pc, residual, out, out_split)
        if (up2 and gemm_x3_enabled() and _UP2_TAPS and residual is None and pc.R == 3 and pc.S == 3 and pc.stride == 1 and pc.pad == 1 and pc.dil == 1
                and pc.cout % 4 == 0 and c_ % 32 == 0):
            split_out = out_split and out is None and pc.cout % 32 == 0      # (the next layer is again a tap-decomposed upsample conv)
            if out is None:
                out = torch.empty((n_, 2 * h_, 2 * w_, pc.cout), dtype=torch.float32, device=x.device)
            outer_tag = _layer_tag
            if _profile is not None and outer_tag is None:
                _layer_tag = (n_, 2 * h_, 2 * w_, pc.cin, pc.cout, pc.R, pc.stride, pc.dil, True, "taps(x3)", 2 * n_ * 4 * h_ * w_ * pc.cout * 9 * pc.cin)
            try:
                _conv_up2_taps(x, pc, out, True, split_out)
            finally:
                _layer_tag = outer_tag
            return SplitRows(out) if split_out else out
        return conv2d(x.float(), pc, residual, out, tile_cfg, split_k, up2, out_split)
    if is16(x):
        return _conv2d16(x, pc, residual, out, up2, tile_cfg, split_k)
    _need_gpu(x, residual, out)
    if out_split and tile_cfg == 0 and split_k == 0 and out is None and _x3_eligible(pc, x.shape[3], residual, up2) and pc.cout % 32 == 0:
        # the consumer takes split rows (e.g. the PSP bottleneck feeding up_1): this conv runs on the LDS-DMA GEMM and writes them
        outer_tag = _layer_tag
        if _profile is not None and outer_tag is None:
            n_, h_, w_, c_ = x.shape
            _layer_tag = (n_, h_, w_, pc.cin, pc.cout, 1, 1, 1, False, "x3(split out)", 2 * n_ * h_ * w_ * pc.cout * pc.cin)
        try:
            return _conv1x1_x3(x, pc, residual, None, True)
        finally:
            _layer_tag = outer_tag
    if _RANGE_GUARD and _math == _lib.MATH_F16X3 and not (float(x.abs().max()) <= _RANGE_LIMIT):      # (NaN compares false)
        prev = set_conv_math("f32")
        try:
            return conv2d(x, pc, residual, out, 0, 0, up2)
        finally:
            set_conv_math(prev)
    x_low = None
    if up2:
        x_low = x
        n_, h_, w_, c_ = x.shape
        if (tile_cfg and tile_cfg not in _PATCH_CFGS) or (not tile_cfg and (split_k or not _AUTOTUNE)):
            x, x_low = resize_nhwc(x, 2 * h_, 2 * w_, _lib.BILINEAR, False), None          # explicit GEMM tile / heuristic plan: materialise
        else:
            # shape carrier; filled only if a plan without a fused upsample is chosen (the patch-resident plans and the Winograd
            # route interpolate while they stage their input)
            x = torch.empty((n_, 2 * h_, 2 * w_, c_), dtype=x.dtype, device="meta")
    dev = x_low.device if x_low is not None else x.device
    N, H, W, Cin = x.shape
    if Cin != pc.cin_pad:
        raise _lib.ArsegError(f"conv expects {pc.cin_pad} input channels (padded), got {Cin}")
    d = ConvDesc()
    in_ld_hi = Cin if x_low is not None else _nhwc_ld(x)
    d.N, d.H, d.W, d.Cin, d.in_ld = N, H, W, Cin, in_ld_hi
    d.Cout = pc.cout
    d.R, d.S, d.stride, d.pad, d.dil = pc.R, pc.S, pc.stride, pc.pad, pc.dil
    d.act, d.prelu_slope = pc.act, pc.slope
    d.tile_cfg, d.split_k = tile_cfg, split_k
    d.math = math = _math
    _arm_range_watch(d, dev)
    w_dev, scale_dev = (pc.w_h3, pc.scale_h3) if math != _lib.MATH_F32 else (pc.w, pc.scale)
    d.out_ld, d.res_ld = pc.cout, pc.cout     # provisional, for the shape query
    lib = _lib.load()
    ho, wo = ctypes.c_int(), ctypes.c_int()
    check(lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)), "conv_out_hw")
    Ho, Wo = ho.value, wo.value
    if out is None:
        out = torch.empty((N, Ho, Wo, pc.cout), dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (N, Ho, Wo, pc.cout):
        raise _lib.ArsegError(f"conv out has shape {tuple(out.shape)}, expected {(N, Ho, Wo, pc.cout)}")
    d.out_ld = _nhwc_ld(out)
    if residual is not None:
        if tuple(residual.shape) != (N, Ho, Wo, pc.cout):
            raise _lib.ArsegError("residual shape mismatch")
        d.res_ld = _nhwc_ld(residual)
    flops = 2 * N * Ho * Wo * pc.cout * pc.R * pc.S * pc.cin

    up_buf = []

    def launch(cfg, sk, record=True):
        xin = x
        d.upsample2x, d.in_ld = 0, in_ld_hi
        if x_low is not None:
            if cfg in _PATCH_CFGS:                              # the patch-resident kernel upsamples while it stages its patch
                xin, d.upsample2x, d.in_ld = x_low, 1, _nhwc_ld(x_low)
            else:                                               # GEMM kernel on an upsampled input: materialise it
                if not up_buf:
                    up_buf.append(torch.empty((N, H, W, Cin), dtype=torch.float32, device=x_low.device))
                xin = resize_nhwc(x_low, H, W, _lib.BILINEAR, False, out=up_buf[0])
        d.tile_cfg, d.split_k = cfg, sk
        nbytes = lib.arseg_conv2d_workspace_bytes(ctypes.byref(d))
        ws = workspace(nbytes, out.device) if nbytes else None
        args = (ctypes.byref(d), _ptr(xin), _ptr(w_dev), _ptr(scale_dev), _ptr(pc.bias), _ptr(residual), _ptr(out), _ptr(ws), nbytes, _stream())
        if record:
            _launch("conv2d", lib.arseg_conv2d_fwd, *args, flops=flops)
        else:
            check(lib.arseg_conv2d_fwd(*args), "conv2d")

    def launch_wino(record=True):
        _conv_wino(x if x_low is None else x_low, pc, residual, out, N, H, W, record, up2=x_low is not None)

    def launch_taps(record=True):
        _conv_up2_taps(x_low, pc, out, record)

    def launch_x3(record=True):
        _conv1x1_x3(x, pc, residual, out, False, None, record)

    def find_native():
        """Plan selection inside the library (arseg_conv2d_find: every candidate timed with HIP events, no Python in the loop).  With a
        fused upsample only the patch-resident plans qualify; None = nothing launched (the Python tuner then tries the rest)."""
        xin = x
        d.upsample2x, d.in_ld = 0, in_ld_hi
        if x_low is not None:
            xin, d.upsample2x, d.in_ld = x_low, 1, _nhwc_ld(x_low)
        nbytes = lib.arseg_conv2d_find_workspace_bytes(ctypes.byref(d))
        ws = workspace(nbytes, out.device) if nbytes else None
        cfg, sk, us = ctypes.c_int(), ctypes.c_int(), ctypes.c_float()
        st = lib.arseg_conv2d_find(ctypes.byref(d), _ptr(xin), _ptr(w_dev), _ptr(scale_dev), _ptr(pc.bias), _ptr(residual), _ptr(out), _ptr(ws),
                                   nbytes, 3, ctypes.byref(cfg), ctypes.byref(sk), ctypes.byref(us), _stream())
        d.upsample2x, d.in_ld = 0, in_ld_hi
        if st > 0:
            check(st, "conv2d_find")                                # a HIP error is not "no plan": raise it
        return (cfg.value, sk.value) if st == _lib.ARSEG_OK else None

    if tile_cfg == 0 and split_k == 0 and _AUTOTUNE:
        key = (dev.index, N, H, W, Cin, pc.cout, pc.R, pc.S, pc.stride, pc.pad, pc.dil, x_low is not None, math)
        wino_ok = getattr(pc, "wino_u", None) is not None and _WINOGRAD
        taps_ok = (x_low is not None and _UP2_TAPS and residual is None and pc.R == 3 and pc.S == 3 and pc.stride == 1 and pc.pad == 1
                   and pc.dil == 1 and pc.cout % 4 == 0)
        x3_ok = _x3_eligible(pc, Cin, residual, x_low is not None) and not wino_ok
        plan = _conv_plans.get(key)
        if (plan == "wino" and not wino_ok) or (plan == "taps" and not taps_ok) or (plan == "x3" and not x3_ok) or plan == "tapsf":      # a persisted plan whose route is switched off / gone: re-tune
            plan = None
        if plan is None:
            plan = find_native() if _NATIVE_FIND else None
            if plan is None:
                plan = _tune_conv(launch, pc, N * Ho * Wo)
            elif x_low is not None:
                # with a fused upsample the library times only the patch-resident plans: also time the GEMM-tile plans on the materialised
                # upsample and keep the faster
                alt = _tune_conv(launch, pc, N * Ho * Wo, allow_patch=False)
                if alt is not None and _time(lambda: launch(*alt, record=False)) < _time(lambda: launch(*plan, record=False)):
                    plan = alt
            if plan is None:
                # nothing could be launched.  The one shape-independent cause is the 2 GiB limit of the kernels' 32-bit buffer
                # offsets on a large batch: split the batch (as creff does) instead of caching a plan that never ran.
                if N > 1 and x_low is None and max(x.numel(), out.numel()) * 4 >= (1 << 31):
                    hN = N // 2
                    conv2d(x[:hN], pc, None if residual is None else residual[:hN], out[:hN])
                    conv2d(x[hN:], pc, None if residual is None else residual[hN:], out[hN:])
                    return out
                launch(0, 0)                                    # raises the library's own error
            if wino_ok:
                try:
                    t_direct = _time(lambda: launch(*plan, record=False))
                    launch_wino(record=False)                   # tunes the batched GEMM underneath
                    if _time(lambda: launch_wino(record=False)) * float(config.conv_wino_margin) < t_direct:
                        plan = "wino"
                except _lib.ArsegError:
                    pass                                        # the Winograd route does not cover this shape: keep the direct plan
            if x3_ok:                                           # split pre-pass + LDS-DMA GEMM against the best implicit-GEMM plan
                try:
                    t_direct = _time(lambda: launch(*plan, record=False))
                    launch_x3(record=False)                     # picks its tile shape
                    if _time(lambda: launch_x3(record=False)) < t_direct:
                        plan = "x3"
                except _lib.ArsegError:
                    pass
            if taps_ok:
                try:
                    t_best = _time((lambda: launch_wino(record=False)) if plan == "wino" else (lambda: launch(*plan, record=False)))
                    launch_taps(record=False)                   # tunes the low-resolution GEMM underneath
                    if _time(lambda: launch_taps(record=False)) < t_best:
                        plan = "taps"
                except _lib.ArsegError:
                    pass
            _conv_plans[key] = plan
        outer_tag = _layer_tag          # the tap-decomposed route calls conv2d for its low-resolution GEMM: the outermost layer keeps the tag
        if _profile is not None and outer_tag is None:
            _layer_tag = (N, H, W, pc.cin, pc.cout, pc.R, pc.stride, pc.dil, x_low is not None, str(plan), flops)
        try:
            if plan == "wino":
                launch_wino()
            elif plan == "taps":
                launch_taps()
            elif plan == "x3":
                launch_x3()
            else:
                launch(*plan)
        finally:
            _layer_tag = outer_tag
    else:
        launch(tile_cfg, split_k)
    return out


def _conv2d16(x, pc, residual, out, up2, tile_cfg=0, split_k=0):
    """conv2d on the 16-bit storage path: one MFMA per product (arseg_conv2d16_fwd), fp32 epilogue (pc.scale / pc.bias)."""
    dt = _need_gpu16(x, residual, out)
    if up2:
        n_, h_, w_, c_ = x.shape
        x = resize_nhwc(x, 2 * h_, 2 * w_, _lib.BILINEAR, False)
    N, H, W, Cin = x.shape
    w16, cin_pad = pc.weights16(x.dtype)
    if Cin != cin_pad:
        raise _lib.ArsegError(f"conv (16-bit) expects {cin_pad} input channels (padded to 8), got {Cin}")
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.in_ld = N, H, W, Cin, _nhwc_ld(x)
    d.Cout = pc.cout
    d.R, d.S, d.stride, d.pad, d.dil = pc.R, pc.S, pc.stride, pc.pad, pc.dil
    d.act, d.prelu_slope = pc.act, pc.slope
    d.tile_cfg = tile_cfg
    d.out_ld, d.res_ld = pc.cout, pc.cout
    lib = _lib.load()
    ho, wo = ctypes.c_int(), ctypes.c_int()
    check(lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)), "conv_out_hw")
    Ho, Wo = ho.value, wo.value
    cout_ld = (pc.cout + 7) // 8 * 8
    if out is None:
        out = torch.empty((N, Ho, Wo, cout_ld), dtype=x.dtype, device=x.device)[..., :pc.cout]
    elif tuple(out.shape) != (N, Ho, Wo, pc.cout):
        raise _lib.ArsegError(f"conv out has shape {tuple(out.shape)}, expected {(N, Ho, Wo, pc.cout)}")
    d.out_ld = _nhwc_ld(out)
    if residual is not None:
        if tuple(residual.shape) != (N, Ho, Wo, pc.cout):
            raise _lib.ArsegError("residual shape mismatch")
        d.res_ld = _nhwc_ld(residual)
    d.split_k = split_k

    def args():
        nbytes = lib.arseg_conv2d16_workspace_bytes(ctypes.byref(d))
        ws = workspace(nbytes, x.device) if nbytes else None
        return (ctypes.byref(d), dt, _ptr(x), _ptr(w16), _ptr(pc.scale), _ptr(pc.bias), _ptr(residual), _ptr(out), _ptr(ws), nbytes, _stream())

    if tile_cfg == 0 and split_k == 0 and _AUTOTUNE:      # per-shape plan: tile / K-step variants x split-K timed once on the device
        key = ("conv16", x.device.index, dt, N, H, W, Cin, pc.cout, pc.R, pc.S, pc.stride, pc.pad, pc.dil)
        plan = _conv_plans.get(key)
        if plan is None:
            best_t = float("inf")
            ktiles = (pc.R * pc.S * Cin + 63) // 64
            for cfg in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):    # 5..8: patch-resident 3x3 plans, 9: the 7x7 stem kernel (EUNSUPPORTED for other shapes)
                for sk in (0, 1, 2, 4, 8):
                    if sk > 1 and (ktiles // sk < 3 or pc.cout % 8 or cfg >= 5):
                        continue
                    if cfg >= 5 and (sk == 1 or (cfg in (6, 8) and pc.cout <= 64)):
                        continue
                    d.tile_cfg, d.split_k = cfg, sk
                    try:
                        t = _time(lambda: check(lib.arseg_conv2d16_fwd(*args()), "conv2d16"))
                    except _lib.ArsegError:
                        continue
                    if t < best_t:
                        plan, best_t = (cfg, sk), t
            _conv_plans[key] = plan = plan or (0, 0)
        d.tile_cfg, d.split_k = plan
    global _layer_tag
    outer_tag, flops16 = _layer_tag, 2 * N * Ho * Wo * pc.cout * pc.R * pc.S * pc.cin
    if _profile is not None and outer_tag is None:      # per-layer table (profile.layers())
        _layer_tag = (N, H, W, pc.cin, pc.cout, pc.R, pc.stride, pc.dil, bool(up2), "16-bit " + str((d.tile_cfg, d.split_k)), flops16)
    try:
        _launch("conv2d", lib.arseg_conv2d16_fwd, *args(), flops=flops16)
    finally:
        _layer_tag = outer_tag
    return out


_WINOGRAD = config.conv_winograd


def _time(fn, reps=6, rounds=2):
    """ms per call: the faster of ``rounds`` averages over ``reps`` calls (plans chosen from one short average were visibly noisy box to box)."""
    fn()
    best = float("inf")
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best


def _conv_wino(x, pc, residual, out, N, H, W, record=True, up2=False):
    """3x3 stride-1 conv as Winograd F(4x4,3x3): input transform -> 36 batched GEMMs -> output transform.  The GEMMs run either on the
    implicit-GEMM kernel in batched 1x1 mode (plan = its tile_cfg) or, with the transformed activations written as split rows, on the
    LDS-DMA kernel of csrc/gemm_x3.hip (plan = 100 + its tile_cfg); whichever was faster when the shape was first seen.
    N,H,W: conv input size; with up2 ``x`` is the half-resolution tensor the input transform upsamples on the fly."""
    lib = _lib.load()
    Cin, Cout, dil = pc.cin_pad, pc.cout, pc.dil
    T = lib.arseg_wino43_tiles(N, H, W, dil)
    V = torch.empty((36, T, Cin), dtype=torch.float32, device=x.device)
    M = torch.empty((36, T, Cout), dtype=torch.float32, device=x.device)
    la = _launch if record else (lambda name, fn, *a, **k: check(fn(*a), name))
    # under f16x3 the transformed activations are stored scaled by 2^-4 (exact; undone in the output transform): B^T d B amplifies by up to
    # 100, and unscaled the split-fp16 operand range would be left for |x| >~ 1.3e3
    vs = 2.0 ** -4 if _math == _lib.MATH_F16X3 else 1.0
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.in_ld = 1, T, 1, Cin, Cin
    d.Cout, d.out_ld, d.res_ld = Cout, Cout, Cout
    d.R, d.S, d.stride, d.pad, d.dil = 1, 1, 1, 0, 1
    d.act, d.prelu_slope = _lib.ACT_NONE, 0.0
    d.batch, d.in_batch_stride, d.w_batch_stride, d.out_batch_stride = 36, T * Cin, Cout * Cin, T * Cout
    d.math = math = _math
    _arm_range_watch(d, x.device)          # the batched GEMM watches the transformed activations it multiplies
    u_dev, scale_dev = (pc.wino_u_h3, pc.wino_scale_h3) if math != _lib.MATH_F32 else (pc.wino_u, pc.scale)
    x3_ok = bool(_GEMM_X3) and math == _lib.MATH_F16X3 and Cin % 32 == 0 and Cout % 4 == 0
    key = ("wino_gemm", x.device.index, T, Cin, Cout, math)
    plan = _conv_plans.get(key)
    if plan is not None and plan >= 100 and not x3_ok:
        plan = None

    def transform(split, la_):
        if split:      # the transform is the last place that sees the GEMM's fp32 operands: it carries the range watch
            la_("wino_input", lib.arseg_wino43_input_split_fwd, _ptr(x), _nhwc_ld(x), _ptr(V), N, H, W, Cin, dil, 1 if up2 else 0, vs,
                ctypes.c_void_p(d.range_flag), 65504.0, _stream())
        else:
            la_("wino_input", lib.arseg_wino43_input_fwd, _ptr(x), _nhwc_ld(x), _ptr(V), N, H, W, Cin, dil, 1 if up2 else 0, vs, _stream())

    def gemm(cfg, rec):
        if cfg >= 100:
            fn, args = lib.arseg_gemm_x3_fwd, (_ptr(V), _ptr(u_dev), _ptr(M), T, Cout, Cin, Cout, 36, T * Cin * 4, Cout * Cin * 4, T * Cout,
                                               _ptr(None), _ptr(None), _ptr(None), 0, _lib.ACT_NONE, 0.0, 0, cfg - 100, _ptr(None), 0.0, _stream())
        else:
            d.tile_cfg, d.split_k = cfg, 1
            fn, args = lib.arseg_conv2d_fwd, (ctypes.byref(d), _ptr(V), _ptr(u_dev), _ptr(None), _ptr(None), _ptr(None), _ptr(M), _ptr(None), 0, _stream())
        if rec:
            _launch("conv2d", fn, *args, flops=2 * 36 * T * Cin * Cout)
        else:
            check(fn(*args), "conv2d(batched)")

    if plan is None:
        quiet = lambda name, fn, *a, **k: check(fn(*a), name)      # noqa: E731
        best, best_t = 0, float("inf")
        transform(False, quiet)
        for cfg in (0, 5, 6, 7, 8, 9, 10, 11, 12) + ((17, 18, 19) if math == _lib.MATH_F16X3 else ()):
            if cfg in (5, 8, 9, 12, 17, 18, 19) and Cout <= 64:
                continue
            if cfg in (18, 19) and Cout <= 128:
                continue
            t = _time(lambda: gemm(cfg, False))
            if t < best_t:
                best, best_t = cfg, t
        if x3_ok:
            transform(True, quiet)
            for cfg in range(100, 107):
                t = _time(lambda: gemm(cfg, False))
                if t < best_t:
                    best, best_t = cfg, t
        plan = _conv_plans[key] = best
    transform(plan >= 100, la)
    gemm(plan, record)
    la("wino_output", lib.arseg_wino43_output_fwd, _ptr(M), _ptr(scale_dev), _ptr(pc.bias), _ptr(residual),
       _nhwc_ld(residual) if residual is not None else 0, _ptr(out), _nhwc_ld(out), N, H, W, Cout, dil, pc.act, pc.slope, 1.0 / vs, _stream())


_UP2_TAPS = config.conv_up2_taps
_GEMM_X3 = config.conv_gemm_x3


def _conv_up2_taps(x_low, pc, out, record=True, out_split=False):
    """conv3x3(pad 1) of the x2 bilinear upsample of ``x_low`` by tap decomposition (csrc/upconv.hip): one 1x1 conv at low resolution
    with the nine taps stacked along the output channels, then the gather that samples the nine planes at the shifted positions of the
    upsampled image and applies the epilogue of ``pc``.  out_split: the gather writes ``out`` (contiguous, Cout % 32 == 0) as split rows."""
    lib = _lib.load()
    n, h, w, _ = x_low.shape
    z = conv2d(x_low, pc.taps())
    if out_split:
        rw = _range_word(out.device) if _RANGE_MODE == "device" else None
        fn, name = lib.arseg_upconv3x3_tap_gather_split_fwd, "up2_tap_gather"
        args = (_ptr(z), 9 * pc.cout, _ptr(pc.scale), _ptr(pc.bias), _ptr(out), n, h, w, pc.cout, pc.act, pc.slope, _ptr(rw), 65504.0, _stream())
    else:
        fn, name = lib.arseg_upconv3x3_tap_gather_fwd, "up2_tap_gather"
        args = (_ptr(z), 9 * pc.cout, _ptr(pc.scale), _ptr(pc.bias), _ptr(out), _nhwc_ld(out), n, h, w, pc.cout, pc.act, pc.slope, _stream())
    if record:
        _launch(name, fn, *args)
    else:
        check(fn(*args), name)


def _tune_conv(launch, pc, m, allow_patch=True):
    ktiles = (pc.R * pc.S * pc.cin_pad + 31) // 32
    best, best_t = (0, 0), float("inf")
    patch_ok = allow_patch and (_math == _lib.MATH_F16X3 and pc.R == 3 and pc.S == 3 and pc.stride == 1 and pc.pad == pc.dil == 1 and pc.cin_pad % 32 == 0)
    for cfg, sk in ([(0, 0)] if allow_patch else []) + _conv_candidates(ktiles, pc.cout, m, patch_ok):
        try:
            launch(cfg, sk, record=False)                          # warm (also sizes the workspace)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                launch(cfg, sk, record=False)
            e.record()
            e.synchronize()
            t = s.elapsed_time(e)
        except _lib.ArsegError:
            continue
        if t < best_t:
            best, best_t = (cfg, sk), t
    return best if best_t < float("inf") else None          # None: no candidate could be launched (nothing to cache)


# ----------------------------------------------------------------------------------------------
# small layers
# ----------------------------------------------------------------------------------------------
def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    if is16(x):
        dt = _need_gpu16(x)
        x = x.contiguous()
        N, H, W, C = x.shape
        out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
        _launch("maxpool", _lib.load().arseg_maxpool3x3s2_16_fwd, _ptr(x), _ptr(out), dt, N, H, W, C, _stream())
        return out
    _need_gpu(x)
    x = x.contiguous()
    N, H, W, C = x.shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
    _launch("maxpool", _lib.load().arseg_maxpool3x3s2_fwd, _ptr(x), _ptr(out), N, H, W, C, _stream())
    return out


def adaptive_avgpool(x: torch.Tensor, oh: int, ow: int, out: Optional[torch.Tensor] = None, out_ld: int = 0, out_n_stride: int = 0
                     ) -> torch.Tensor:
    """NHWC -> [N,oh,ow,C]; or, with ``out`` (a base tensor/view whose data_ptr is the first bin of image 0), into rows of a
    wider matrix: element (n, bin, c) at out + n*out_n_stride + bin*out_ld + c."""
    _need_gpu(x, out)
    N, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, oh, ow, C), dtype=torch.float32, device=x.device)
    _launch("adaptive_avgpool", _lib.load().arseg_adaptive_avgpool_fwd, _ptr(x), _nhwc_ld(x), _ptr(out), out_ld, out_n_stride, N, H, W, C,
            oh, ow, _stream())
    return out


def psp_pool_matrix(x: torch.Tensor, sizes) -> torch.Tensor:
    """The folded pyramid's block-structured pooled matrix [N, sum(s^2), 1, len(sizes)*C]: level i's adaptive average pool in columns
    [i*C, (i+1)*C) of its s_i^2 rows, zeros elsewhere -- written entirely by the pooling launches (no fill)."""
    _need_gpu(x)
    N, H, W, C = x.shape
    n, rows = len(sizes), sum(s * s for s in sizes)
    out = torch.empty((N, rows, 1, n * C), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    arr = (ctypes.c_int * n)(*[int(s) for s in sizes])
    nb = lib.arseg_psp_pool_matrix_workspace_bytes(N, H, W, C, n, arr) if (n <= 4 and C % 4 == 0 and N <= 65535) else 0
    if nb:          # one pass over the map: the cells of the grid spanned by all bin edges are summed once, then combined per bin
        ws = torch.empty((nb // 4,), dtype=torch.float32, device=x.device)
        _launch("adaptive_avgpool", lib.arseg_psp_pool_matrix_fwd, _ptr(x), _nhwc_ld(x), _ptr(out), _ptr(ws), nb, N, H, W, C, n, arr, _stream())
        return out
    off = 0
    for i, s in enumerate(sizes):
        _launch("adaptive_avgpool", _lib.load().arseg_adaptive_avgpool_blockrow_fwd, _ptr(x), _nhwc_ld(x), _ptr(out[0, off]), rows * n * C,
                N, H, W, C, s, s, n, i, _stream())
        off += s * s
    return out


_psp_interp = {}


def psp_x3_foldable(pc) -> bool:
    """psp_bottleneck_x3 pre-divides the pyramid terms by the epilogue's per-channel scale: a channel whose folded scale is 0 (pruned /
    zero-initialised gamma) or tiny would give inf / overflow the split range, so such a module stays on the prior-sum + residual path.
    Decided once per packed conv (one host read at the first forward), cached."""
    ok = pc.__dict__.get("_x3_foldable")
    if ok is None:
        sc = pc.scale_h3.detach().abs()
        ok = pc.__dict__["_x3_foldable"] = bool(torch.isfinite(sc).all().item()) and float(sc.min().item()) > 1e-4 * max(float(sc.max().item()), 1e-30)
    return ok


def psp_bottleneck_x3(feats: torch.Tensor, t: torch.Tensor, pc, sizes, out_split: bool = True):
    """PSPModule's folded bottleneck (model/pspnet.py:14-31) as ONE GEMM:  relu(W_f f + b + sum_s upsample(t_s))  with the pyramid sum written as
    B . T -- B [H*W, 64] the bilinear interpolation matrix of the pooled rows (built once per shape by running psp_prior_sum on an identity, so it
    holds exactly the weights that kernel applies), T the per-image pyramid terms -- and concatenated along K of the bottleneck GEMM
    (arseg_gemm_x3_cat_fwd): [f | B] . [W_f ; T]^T.  The 92 MB prior tensor (written by one kernel, read back as a residual by the next) is
    gone: two more K steps.  feats [N,H,W,C] fp32, t [N, rows, C_out] (rows = sum s^2 <= 64); returns SplitRows / tensor [N,H,W,C_out]."""
    lib = _lib.load()
    N, H, W, C = feats.shape
    rows, Cout, dev = t.shape[1], pc.cout, feats.device
    key = (dev.index, H, W, tuple(sizes))
    B = _psp_interp.get(key)
    if B is None:
        eye = torch.zeros((1, rows, 64), dtype=torch.float32, device=dev)
        eye[0, torch.arange(rows), torch.arange(rows)] = 1.0
        B = _psp_interp[key] = split_rows(psp_prior_sum(eye, sizes, H, W))          # [1,H,W,64] split rows, shared by the batch
    fac = pc.__dict__.get("_x3_unscale")
    if fac is None:
        fac = pc.__dict__["_x3_unscale"] = (1.0 / pc.scale_h3).contiguous()          # the epilogue multiplies the accumulator by scale_h3
        # (a channel whose folded scale is 0 or tiny cannot be un-scaled: psp_x3_foldable() keeps such a module on the residual path)
    # the per-image pyramid operand [N, Cout, 64] as split rows, straight from t (no torch arithmetic inside the step)
    w2s = SplitRows(torch.empty((N, Cout, 1, 64), dtype=torch.float32, device=dev))
    _launch("psp_w2_split", lib.arseg_psp_w2_split_fwd, _ptr(t.contiguous()), _ptr(fac), _ptr(w2s.t), N, rows, Cout, _stream())
    xs = split_rows(feats)
    out = torch.empty((N, H, W, Cout), dtype=torch.float32, device=dev)
    rw = _range_word(dev) if (out_split and _RANGE_MODE == "device") else None

    def run(c, rec):
        args = (_ptr(xs.t), _ptr(pc.w_h3), _ptr(B.t), _ptr(w2s.t), _ptr(out), H * W, Cout, C, 64, Cout, N, H * W * C * 4, 0, 0, Cout * 64 * 4, H * W * Cout,
                _ptr(pc.scale_h3), _ptr(pc.bias), pc.act, pc.slope, 1 if out_split else 0, c, _ptr(rw), 65504.0, _stream())
        if rec:
            _launch("conv2d", lib.arseg_gemm_x3_cat_fwd, *args, flops=2 * N * H * W * (C + 64) * Cout)
        else:
            check(lib.arseg_gemm_x3_cat_fwd(*args), "gemm_x3_cat")

    pkey = ("x3cat", dev.index, N, H * W, C, Cout, bool(out_split))
    cfg = _conv_plans.get(pkey)
    if cfg is None:
        best_t = float("inf")
        for c in range(7):
            tm = _time(lambda: run(c, False))
            if tm < best_t:
                cfg, best_t = c, tm
        _conv_plans[pkey] = cfg
    global _layer_tag
    outer = _layer_tag
    if _profile is not None and outer is None:
        _layer_tag = (N, H, W, C, Cout, 1, 1, 1, False, "x3(cat, split out)", 2 * N * H * W * C * Cout)
    try:
        run(cfg, True)
    finally:
        _layer_tag = outer
    return SplitRows(out) if out_split else out


def psp_prior_sum(t: torch.Tensor, sizes, H: int, W: int) -> torch.Tensor:
    """t [N, sum(s^2), C] (per-level maps after the folded 1x1 convs) -> [N,H,W,C] sum of bilinear upsamples."""
    _need_gpu(t)
    t = t.contiguous()
    N, rows, C = t.shape
    if rows != sum(s * s for s in sizes):
        raise _lib.ArsegError("psp_prior_sum: row count does not match the pyramid sizes")
    out = torch.empty((N, H, W, C), dtype=torch.float32, device=t.device)
    arr = (ctypes.c_int * len(sizes))(*[int(s) for s in sizes])
    _launch("psp_prior_sum", _lib.load().arseg_psp_prior_sum_fwd, _ptr(t), _ptr(out), N, H, W, C, len(sizes), arr, _stream())
    return out


def global_reduce(x: torch.Tensor, op: int) -> torch.Tensor:
    """NHWC -> [N,1,1,C] mean or max over (H,W)."""
    if is16(x):
        dt = _need_gpu16(x)
        if op != _lib.REDUCE_MEAN:
            raise _lib.ArsegError("16-bit path: only the mean reduction is built (BiSeNet ARM / FFM / conv_avg)")
        N, H, W, C = x.shape
        out = torch.empty((N, 1, 1, C), dtype=x.dtype, device=x.device)
        nb = _lib.load().arseg_global_mean16_workspace_bytes(N, H, W, C)
        ws = workspace(nb, x.device)
        _launch("global_reduce", _lib.load().arseg_global_mean16_fwd, _ptr(x), _nhwc_ld(x), _ptr(out), dt, N, H, W, C, _ptr(ws), nb, _stream())
        return out
    _need_gpu(x)
    N, H, W, C = x.shape
    out = torch.empty((N, 1, 1, C), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nb = lib.arseg_global_reduce_workspace_bytes(N, H, W, C)
    ws = torch.empty((nb // 4,), dtype=torch.float32, device=x.device) if nb else None       # (large map, few images: two-stage reduce)
    _launch("global_reduce", lib.arseg_global_reduce_ws_fwd, _ptr(x), _nhwc_ld(x), _ptr(out), _ptr(ws), nb, N, H, W, C, op, _stream())
    return out


def resize_nhwc(x: torch.Tensor, Hout: int, Wout: int, mode: int, align_corners: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if is16(x):
        dt = _need_gpu16(x, out)
        N, H, W, C = x.shape
        if out is None:
            out = torch.empty((N, Hout, Wout, C), dtype=x.dtype, device=x.device)
        _launch("resize_nhwc", _lib.load().arseg_resize16_fwd, _ptr(x), _ptr(out), dt, N, C, H, W, Hout, Wout, mode, 1 if align_corners else 0,
                _nhwc_ld(x), _nhwc_ld(out), _stream())
        return out
    _need_gpu(x, out)
    N, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, Hout, Wout, C), dtype=torch.float32, device=x.device)
    _launch("resize_nhwc", _lib.load().arseg_resize_fwd, _ptr(x), _ptr(out), N, C, H, W, Hout, Wout, mode, 1 if align_corners else 0, _lib.NHWC,
                                       _nhwc_ld(x), _nhwc_ld(out), _stream())
    return out


def resize_nchw(x: torch.Tensor, Hout: int, Wout: int, mode: int, align_corners: bool) -> torch.Tensor:
    _need_gpu(x)
    x = x.contiguous()
    N, C, H, W = x.shape
    out = torch.empty((N, C, Hout, Wout), dtype=torch.float32, device=x.device)
    _launch("resize_nchw", _lib.load().arseg_resize_fwd, _ptr(x), _ptr(out), N, C, H, W, Hout, Wout, mode, 1 if align_corners else 0, _lib.NCHW, 0, 0,
                                       _stream())
    return out


def scale_add(x: torch.Tensor, scale: torch.Tensor, add_full: Optional[torch.Tensor] = None, add_vec: Optional[torch.Tensor] = None
              ) -> torch.Tensor:
    """out = x * scale[n,c] (+ add_full[n,h,w,c]) (+ add_vec[n,c]); x NHWC contiguous, scale/add_vec [N,1,1,C]."""
    if is16(x):
        dt = _need_gpu16(x, scale, add_full, add_vec)
        x = x.contiguous()
        N, H, W, C = x.shape
        out = torch.empty_like(x)
        _launch("scale_add", _lib.load().arseg_scale_add16_fwd, _ptr(x), _ptr(scale.contiguous()), _ptr(None if add_full is None else add_full.contiguous()),
                _ptr(None if add_vec is None else add_vec.contiguous()), _ptr(out), dt, N, H * W, C, _stream())
        return out
    _need_gpu(x, scale, add_full, add_vec)
    x = x.contiguous()
    N, H, W, C = x.shape
    out = torch.empty_like(x)
    if add_full is not None:
        add_full = add_full.contiguous()
    _launch("scale_add", _lib.load().arseg_scale_add_fwd, _ptr(x), _ptr(scale.contiguous()), _ptr(add_full),
                                          _ptr(None if add_vec is None else add_vec.contiguous()), _ptr(out), N, H * W, C, _stream())
    return out


def head(p_nhwc: torch.Tensor, wf: torch.Tensor, bf: torch.Tensor, log_softmax: bool) -> torch.Tensor:
    """1x1 classifier on an NHWC feature -> NCHW logits (optionally LogSoftmax over classes)."""
    if is16(p_nhwc):
        dt = _need_gpu16(p_nhwc)
        _need_gpu(wf, bf)
        N, H, W, C = p_nhwc.shape
        n_cls = wf.shape[0]
        out = torch.empty((N, n_cls, H, W), dtype=torch.float32, device=p_nhwc.device)
        _launch("head", _lib.load().arseg_head16_fwd, _ptr(p_nhwc), _nhwc_ld(p_nhwc), dt, _ptr(wf), _ptr(bf), _ptr(out), N, H * W, C, n_cls,
                1 if log_softmax else 0, _stream())
        return out
    _need_gpu(p_nhwc, wf, bf)
    N, H, W, C = p_nhwc.shape
    n_cls = wf.shape[0]
    out = torch.empty((N, n_cls, H, W), dtype=torch.float32, device=p_nhwc.device)
    _launch("head", _lib.load().arseg_head_fwd, _ptr(p_nhwc), _nhwc_ld(p_nhwc), _ptr(wf), _ptr(bf), _ptr(out), N, H * W, C, n_cls,
                                     1 if log_softmax else 0, _stream())
    return out


def frame_to_nhwc4(img: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """NCHW RGB frame -> NHWC4, bilinear(align_corners=True) resized to (h,w) (evaluation.py:186-188)."""
    _need_gpu(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    if C != 3:
        raise _lib.ArsegError("frame_to_nhwc4 expects 3 input channels")
    out = torch.empty((N, h, w, 4), dtype=torch.float32, device=img.device)
    _launch("frame_to_nhwc4", _lib.load().arseg_frame_to_nhwc4_fwd, _ptr(img), _ptr(out), N, H, W, h, w, _stream())
    return out


def frame_u8_to_nhwc4(img_u8: torch.Tensor, h: int, w: int, mean, std) -> torch.Tensor:
    """Decoded uint8 frames [N,H,W,3] (HWC, on the GPU) -> normalised NHWC4 [N,h,w,4] (ToTensor + Normalize + bilinear
    align_corners=True downscale in one kernel; the float frame is never materialised)."""
    if img_u8.dtype != torch.uint8 or not img_u8.is_cuda or img_u8.dim() != 4 or img_u8.shape[-1] != 3:
        raise _lib.ArsegError("frame_u8_to_nhwc4 expects a CUDA uint8 tensor [N,H,W,3]")
    img_u8 = img_u8.contiguous()
    N, H, W, _ = img_u8.shape
    out = torch.empty((N, h, w, 4), dtype=torch.float32, device=img_u8.device)
    m3, s3 = (ctypes.c_float * 3)(*[float(v) for v in mean]), (ctypes.c_float * 3)(*[float(v) for v in std])
    _launch("frame_u8_to_nhwc4", _lib.load().arseg_frame_u8_to_nhwc4_fwd, _ptr(img_u8), _ptr(out), N, H, W, h, w, m3, s3, _stream())
    return out


def merge_motion(flows: torch.Tensor, frame_start: int = 0) -> torch.Tensor:
    """Codec motion fields int16 [F+1,H,W,3] (mv_x, mv_y quarter-pel, reference index; on the GPU) -> accumulated quarter-pel
    motion to the keyframe, int16 [F+1,H,W,2] (frame 0 = -1, as the reference's mergeMotion leaves it)."""
    if flows.dtype != torch.int16 or not flows.is_cuda or flows.dim() != 4 or flows.shape[-1] != 3:
        raise _lib.ArsegError("merge_motion expects a CUDA int16 tensor [F+1,H,W,3]")
    flows = flows.contiguous()
    F1, H, W, _ = flows.shape
    lib = _lib.load()
    nbytes = lib.arseg_merge_motion_workspace_bytes(F1 - 1, H, W)
    ws = workspace(nbytes, flows.device)
    out = torch.empty((F1, H, W, 2), dtype=torch.int16, device=flows.device)
    _launch("merge_motion", lib.arseg_merge_motion_fwd, _ptr(flows), _ptr(out), _ptr(ws), nbytes, F1 - 1, frame_start, H, W, _stream())
    return out


def argmax_confusion(logits: torch.Tensor, label: Optional[torch.Tensor], H: int, W: int, hist: Optional[torch.Tensor] = None,
                     ignore_label: int = 255, want_pred: bool = True, align_corners: bool = True):
    """Evaluator tail (evaluation.py:201-209): returns (pred int32 [N,H,W] or None, hist int64 [n_cls,n_cls] or None).
    ``align_corners=False``: the resize is BiSeNetOutput's ``nn.Upsample(x8, align_corners=False)`` (model/bisenet.py:215-216) --
    head logits at 1/8 resolution go straight to the argmax, the full-resolution logits are never written."""
    _need_gpu(logits)
    logits = logits.contiguous()
    N, n_cls, h, w = logits.shape
    pred = torch.empty((N, H, W), dtype=torch.int32, device=logits.device) if want_pred else None
    if label is not None:
        _need_gpu(label, dtype=torch.int64)
        label = label.contiguous()
        if hist is None:
            hist = torch.zeros((n_cls, n_cls), dtype=torch.int64, device=logits.device)
    _launch("argmax_confusion", _lib.load().arseg_argmax_confusion_fwd, _ptr(logits), _ptr(label), _ptr(pred), _ptr(hist if label is not None else None), N,
                                                 n_cls, h, w, H, W, ignore_label, 1 if align_corners else 0, _stream())
    return pred, hist


def _apply_config():
    """Push ``ops.config`` into the module-level switches the hot paths read."""
    global _AUTOTUNE, _math, _RANGE_MODE, _RANGE_GUARD, _NATIVE_FIND, _WINOGRAD, _UP2_TAPS, _PLAN_FILE
    if config.conv_plan_file != _PLAN_FILE:          # a new plan file: its plans join the cache, later plans are mirrored to it
        _PLAN_FILE = config.conv_plan_file
        _conv_plans.load()
    _AUTOTUNE, _math = config.conv_autotune, _MATH_NAMES[config.conv_math]
    _RANGE_MODE = config.conv_range_guard if config.conv_range_guard in ("device", "host", "off") else "device"
    _RANGE_GUARD = _RANGE_MODE == "host"
    global _GEMM_X3
    _NATIVE_FIND, _WINOGRAD, _UP2_TAPS = config.conv_find != "python", config.conv_winograd, config.conv_up2_taps
    _GEMM_X3 = config.conv_gemm_x3
