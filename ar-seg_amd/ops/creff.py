"""Motion-vector warp and CReFF wrappers: arseg_warp_*, arseg_creff_*."""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from .. import _lib
from ._base import _need_gpu, _need_gpu16, _ptr, _stream, is16
from ._config import config
from ._profile import launch
from .layers import cast, from_c8


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
    launch("warp", _lib.load().arseg_warp_fwd, _ptr(feature), _ptr(flow), fd, _ptr(out), N, C, H, W, layout, out_layout, _stream())
    return out


def mv_resize(mv_q: torch.Tensor, Hp: int, Wp: int) -> torch.Tensor:
    """int16 quarter-pel [N,H,W,2] -> float64 [N,Hp,Wp,2] (evaluation.py:176-180)."""
    _need_gpu(mv_q, dtype=torch.int16)
    mv_q = mv_q.contiguous()
    N, H, W, _ = mv_q.shape
    out = torch.empty((N, Hp, Wp, 2), dtype=torch.float64, device=mv_q.device)
    launch("mv_resize", _lib.load().arseg_mv_resize_fwd, _ptr(mv_q), _ptr(out), N, H, W, Hp, Wp, _stream())
    return out


def flow_resize(flow: torch.Tensor, Hp: int, Wp: int) -> torch.Tensor:
    """float flow [N,H,W,2] in pixels (fp32 / fp64) -> float64 [N,Hp,Wp,2] (evaluation.py:176-180), any values."""
    _need_gpu(flow, dtype=None)
    if flow.dtype not in (torch.float32, torch.float64):
        raise _lib.ArsegError(f"flow must be float32 or float64, got {flow.dtype}")
    flow = flow.contiguous()
    N, H, W, _ = flow.shape
    out = torch.empty((N, Hp, Wp, 2), dtype=torch.float64, device=flow.device)
    launch("flow_resize", _lib.load().arseg_flow_resize_fwd, _ptr(flow), _lib.FLOW_F64 if flow.dtype == torch.float64 else _lib.FLOW_F32, _ptr(out),
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
    launch("warp_mvq", _lib.load().arseg_warp_mvq_fwd, _ptr(feature_nhwc), _ptr(mv_q), _ptr(out), N, C, Hp, Wp, H, W, out_layout, _stream(),
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
    p_out = torch.empty_like(hr_c8)
    logits, wf, bf, n_cls = None, None, None, 0
    if head is not None:
        wf, bf = head
        n_cls = wf.shape[0]
        logits = torch.empty((N, n_cls, Hp, Wp), dtype=torch.float32, device=hr_c8.device)
    # kernel choice: explicit arguments of the ABI; the knobs (A/B measurements, tests) live in ops.config, not in the library
    impl = {"mfma": 1, "valu": 2}.get(config.creff_impl, 0)
    tile_rows = config.creff_tile_rows
    per = max(1, min(N, ((1 << 31) - 1) // (C * Hp * Wp * 4)))          # the kernels address p_out with 32-bit buffer offsets over the launch's batch
    for i in range(0, N, per):          # (normally one launch; every launch writes its slice of the one output)
        b = min(per, N - i)
        launch("creff", _lib.load().arseg_creff_fwd_ex, _ptr(hr_c8[i:i + b]), _ptr(lr_nhwc[i:i + b]), _ptr(attn.wq), _ptr(attn.bq), _ptr(attn.wk), _ptr(attn.bk),
               _ptr(attn.wv), _ptr(attn.bv), _ptr(p_out[i:i + b]), _ptr(wf), _ptr(bf), n_cls, _ptr(None if logits is None else logits[i:i + b]),
               1 if log_softmax else 0, b, C, Hp, Wp, hp, wp, kH, kW, impl, tile_rows if tile_rows in (8, 16) else 0, _stream(),
               flops=b * Hp * Wp * C * (250 + 2 * n_cls),
               nbytes=4 * b * (2 * C * Hp * Wp + C * hp * wp + n_cls * Hp * Wp))
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
            launch("warp_mvq", _lib.load().arseg_warp_mvq16_shared_fwd, _ptr(refs16[0]), 0, dt, _ptr(mv_q), _ptr(ref_c8), B, C, Hp, Wp,
                    H, W, _stream(), nbytes=B * (2 * C * Hp * Wp + 4 * C * Hp * Wp + 4 * H * W))
        else:
            for b in range(B):
                launch("warp_mvq", _lib.load().arseg_warp_mvq16_fwd, _ptr(refs16[b]), dt, _ptr(mv_q[b:b + 1]), _ptr(ref_c8[b:b + 1]), 1, C, Hp, Wp,
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
    n_cls = 0 if head is None else head[0].shape[0]
    per = _warp_frames_per_launch(B, C, Hp, Wp, hp, wp, n_cls, kH, kW)
    shape = (B, C // 8, Hp, Wp, 8) if p_layout == _lib.C8 else (B, Hp, Wp, C)
    p_out = torch.empty(shape, dtype=torch.float32, device=lr_nhwc.device)
    logits, wf, bf = None, None, None
    if head is not None:
        wf, bf = head
        logits = torch.empty((B, n_cls, Hp, Wp), dtype=torch.float32, device=lr_nhwc.device)
    impl = {"tiles": 1, "roll": 2}.get(config.creff_warp_impl, 0)
    for i in range(0, B, per):          # one launch unless the kernel's addressing caps it; every launch writes its slice of the one output
        b = min(per, B - i)
        ptrs = (ctypes.c_void_p * b)(*[r.data_ptr() for r in refs[i:i + b]])
        launch("creff_warp", _lib.load().arseg_creff_warp_fwd_ex, ptrs, _ptr(mv_q[i:i + b]), H, W, _ptr(lr_nhwc[i:i + b]), _ptr(attn.wq), _ptr(attn.bq),
                _ptr(attn.wk), _ptr(attn.bk), _ptr(attn.wv), _ptr(attn.bv), _ptr(p_out[i:i + b]), p_layout, _ptr(wf), _ptr(bf), n_cls,
                _ptr(None if logits is None else logits[i:i + b]), 1 if log_softmax else 0, b, C, Hp, Wp, hp, wp, kH, kW, impl,
                max(0, int(config.creff_seg_rows)), max(0, int(config.creff_max_wgs)), _stream(),
                flops=b * Hp * Wp * C * (250 + 2 * n_cls),
                nbytes=b * (4 * (2 * C * Hp * Wp + C * hp * wp + n_cls * Hp * Wp) + 4 * H * W))
    return p_out, logits


def _warp_frames_per_launch(B, C, Hp, Wp, hp, wp, n_cls, kH=7, kW=7) -> int:
    """Frames one arseg_creff_warp_fwd_ex launch takes for this shape under the current knobs: 32 pointers; the rolling kernel addresses every
    frame through its own buffer descriptor (any batch), the tile kernel the whole batch through one (N x C x Hp x Wp x 4 < 2 GiB) -- asked of
    the library's own dispatch query, not restated here."""
    impl = {"tiles": 1, "roll": 2}.get(config.creff_warp_impl, 0)
    sel = _lib.load().arseg_creff_warp_select
    args = (C, Hp, Wp, hp, wp, kH, kW, n_cls, impl, max(0, int(config.creff_seg_rows)), max(0, int(config.creff_max_wgs)))
    per = min(B, 32)
    if per > 1 and sel(per, *args) == _lib.ARSEG_EUNSUPPORTED:          # not as one launch: what the batch-wide descriptors admit
        per = max(1, min(per, ((1 << 31) - 1) // (C * Hp * Wp * 4)))
    return per


def creff_warp_kernel(B: int, C: int, Hp: int, Wp: int, hp: int, wp: int, n_cls: int, kH: int = 7, kW: int = 7) -> str:
    """Which kernel ``creff_warp`` runs for this launch under the current knobs: "roll" (csrc/creff_roll.hip), "tiles" (csrc/creff_rr.hip) or
    "two-kernel" (arseg_warp_mvq_fwd + arseg_creff_fwd: shapes the fused entry point does not cover).  A pure query of the library's own
    dispatch rule (arseg_creff_warp_select) -- bench.py labels its roofline line with it instead of restating the rule."""
    impl = {"tiles": 1, "roll": 2}.get(config.creff_warp_impl, 0)
    per = _warp_frames_per_launch(B, C, Hp, Wp, hp, wp, n_cls, kH, kW)
    st = _lib.load().arseg_creff_warp_select(min(B, per), C, Hp, Wp, hp, wp, kH, kW, n_cls, impl, max(0, int(config.creff_seg_rows)),
                                             max(0, int(config.creff_max_wgs)))
    if st == _lib.ARSEG_EUNSUPPORTED and impl != 2:
        return "two-kernel"
    _lib.check(st if st < 0 else 0, "creff_warp_select")
    return {1: "tiles", 2: "roll"}[st]
