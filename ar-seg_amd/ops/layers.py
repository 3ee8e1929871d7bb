"""Layout helpers, the localAttention pair, ingest, and the small layers (pooling, resizes, ARM / FFM scaling, heads, evaluator tail,
mergeMotion): thin wrappers, one ABI call each."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _lib
from ._base import _DT16, _need_gpu, _need_gpu16, _nhwc_ld, _ptr, _stream, is16, workspace
from ._profile import launch


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 <-> fp16 / bf16 element conversion on the GPU (round to nearest even), any shape with numel % 8 == 0."""
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    code = lambda d: _lib.DT_F32 if d == torch.float32 else _DT16[d]
    launch("cast", _lib.load().arseg_cast_fwd, _ptr(x), code(x.dtype), _ptr(out), code(dtype), x.numel(), _stream())
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
    launch("frame_to_nhwc8", _lib.load().arseg_frame_to_nhwc8_16_fwd, _ptr(img), _ptr(out), _DT16[dtype], N, H, W, h, w, _stream())
    return out


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
    launch("nchw_to_nhwc", _lib.load().arseg_nchw_to_nhwc_fwd, _ptr(x), _ptr(out), N, C, H * W, C, _stream())
    return out


def as_nchw(x_nhwc: torch.Tensor) -> torch.Tensor:
    """Physical NHWC [N,H,W,C] -> logical NCHW view (channels_last strides, no copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


def to_nchw_contiguous(x_nhwc: torch.Tensor) -> torch.Tensor:
    _need_gpu(x_nhwc)
    N, H, W, C = x_nhwc.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x_nhwc.device)
    launch("nhwc_to_nchw", _lib.load().arseg_nhwc_to_nchw_fwd, _ptr(x_nhwc), C, _ptr(out), N, C, H * W, _stream())
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
    launch("to_c8", _lib.load().arseg_to_c8_fwd, _ptr(x), layout, ld, _ptr(out), N, C, H * W, _stream())
    return out


def from_c8(x: torch.Tensor, layout: int) -> torch.Tensor:
    _need_gpu(x)
    N, CB, H, W, _ = x.shape
    C = CB * 8
    shape = (N, C, H, W) if layout == _lib.NCHW else (N, H, W, C)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    launch("from_c8", _lib.load().arseg_from_c8_fwd, _ptr(x), _ptr(out), layout, C, N, C, H * W, _stream())
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
        launch("local_similar", _lib.load().arseg_local_similar_nhwc_fwd, _ptr(q), _ptr(k), C, _ptr(out), N, C, H, W, kH, kW, _stream())
        return out
    q, k = q.contiguous(), k.contiguous()
    launch("local_similar", _lib.load().arseg_local_similar_fwd, _ptr(q), _ptr(k), _ptr(out), N, C, H, W, kH, kW, _stream())
    return out


def local_weighting(v: torch.Tensor, w: torch.Tensor, kH: int, kW: int) -> torch.Tensor:
    """localAttention.weighting_forward; a channels_last ``v`` gives a channels_last result through the NHWC variant."""
    _need_gpu(v, w)
    w = w.contiguous()
    N, C, H, W = v.shape
    if _is_cl(v):
        out = torch.empty_like(v, memory_format=torch.channels_last)
        launch("local_weighting", _lib.load().arseg_local_weighting_nhwc_fwd, _ptr(v), _ptr(w), C, _ptr(out), N, C, H, W, kH, kW, _stream())
        return out
    v = v.contiguous()
    out = torch.empty_like(v)
    launch("local_weighting", _lib.load().arseg_local_weighting_fwd, _ptr(v), _ptr(w), _ptr(out), N, C, H, W, kH, kW, _stream())
    return out


# ----------------------------------------------------------------------------------------------
# small layers
# ----------------------------------------------------------------------------------------------
def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    if is16(x):
        dt = _need_gpu16(x)
        x = x.contiguous()
        N, H, W, C = x.shape
        out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
        launch("maxpool", _lib.load().arseg_maxpool3x3s2_16_fwd, _ptr(x), _ptr(out), dt, N, H, W, C, _stream())
        return out
    _need_gpu(x)
    x = x.contiguous()
    N, H, W, C = x.shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
    launch("maxpool", _lib.load().arseg_maxpool3x3s2_fwd, _ptr(x), _ptr(out), N, H, W, C, _stream())
    return out


def adaptive_avgpool(x: torch.Tensor, oh: int, ow: int, out: Optional[torch.Tensor] = None, out_ld: int = 0, out_n_stride: int = 0
                     ) -> torch.Tensor:
    """NHWC -> [N,oh,ow,C]; or, with ``out`` (a base tensor/view whose data_ptr is the first bin of image 0), into rows of a
    wider matrix: element (n, bin, c) at out + n*out_n_stride + bin*out_ld + c."""
    _need_gpu(x, out)
    N, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, oh, ow, C), dtype=torch.float32, device=x.device)
    launch("adaptive_avgpool", _lib.load().arseg_adaptive_avgpool_fwd, _ptr(x), _nhwc_ld(x), _ptr(out), out_ld, out_n_stride, N, H, W, C,
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
        launch("adaptive_avgpool", lib.arseg_psp_pool_matrix_fwd, _ptr(x), _nhwc_ld(x), _ptr(out), _ptr(ws), nb, N, H, W, C, n, arr, _stream())
        return out
    off = 0
    for i, s in enumerate(sizes):
        launch("adaptive_avgpool", _lib.load().arseg_adaptive_avgpool_blockrow_fwd, _ptr(x), _nhwc_ld(x), _ptr(out[0, off]), rows * n * C,
                N, H, W, C, s, s, n, i, _stream())
        off += s * s
    return out


def psp_prior_sum(t: torch.Tensor, sizes, H: int, W: int) -> torch.Tensor:
    """t [N, sum(s^2), C] (per-level maps after the folded 1x1 convs) -> [N,H,W,C] sum of bilinear upsamples."""
    _need_gpu(t)
    t = t.contiguous()
    N, rows, C = t.shape
    if rows != sum(s * s for s in sizes):
        raise _lib.ArsegError("psp_prior_sum: row count does not match the pyramid sizes")
    out = torch.empty((N, H, W, C), dtype=torch.float32, device=t.device)
    arr = (ctypes.c_int * len(sizes))(*[int(s) for s in sizes])
    launch("psp_prior_sum", _lib.load().arseg_psp_prior_sum_fwd, _ptr(t), _ptr(out), N, H, W, C, len(sizes), arr, _stream())
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
        launch("global_reduce", _lib.load().arseg_global_mean16_fwd, _ptr(x), _nhwc_ld(x), _ptr(out), dt, N, H, W, C, _ptr(ws), nb, _stream())
        return out
    _need_gpu(x)
    N, H, W, C = x.shape
    out = torch.empty((N, 1, 1, C), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nb = lib.arseg_global_reduce_workspace_bytes(N, H, W, C)
    ws = torch.empty((nb // 4,), dtype=torch.float32, device=x.device) if nb else None       # (large map, few images: two-stage reduce)
    launch("global_reduce", lib.arseg_global_reduce_ws_fwd, _ptr(x), _nhwc_ld(x), _ptr(out), _ptr(ws), nb, N, H, W, C, op, _stream())
    return out


def resize_nhwc(x: torch.Tensor, Hout: int, Wout: int, mode: int, align_corners: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if is16(x):
        dt = _need_gpu16(x, out)
        N, H, W, C = x.shape
        if out is None:
            out = torch.empty((N, Hout, Wout, C), dtype=x.dtype, device=x.device)
        launch("resize_nhwc", _lib.load().arseg_resize16_fwd, _ptr(x), _ptr(out), dt, N, C, H, W, Hout, Wout, mode, 1 if align_corners else 0,
                _nhwc_ld(x), _nhwc_ld(out), _stream())
        return out
    _need_gpu(x, out)
    N, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, Hout, Wout, C), dtype=torch.float32, device=x.device)
    launch("resize_nhwc", _lib.load().arseg_resize_fwd, _ptr(x), _ptr(out), N, C, H, W, Hout, Wout, mode, 1 if align_corners else 0, _lib.NHWC,
                                       _nhwc_ld(x), _nhwc_ld(out), _stream())
    return out


def resize_nchw(x: torch.Tensor, Hout: int, Wout: int, mode: int, align_corners: bool) -> torch.Tensor:
    _need_gpu(x)
    x = x.contiguous()
    N, C, H, W = x.shape
    out = torch.empty((N, C, Hout, Wout), dtype=torch.float32, device=x.device)
    launch("resize_nchw", _lib.load().arseg_resize_fwd, _ptr(x), _ptr(out), N, C, H, W, Hout, Wout, mode, 1 if align_corners else 0, _lib.NCHW, 0, 0,
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
        launch("scale_add", _lib.load().arseg_scale_add16_fwd, _ptr(x), _ptr(scale.contiguous()), _ptr(None if add_full is None else add_full.contiguous()),
                _ptr(None if add_vec is None else add_vec.contiguous()), _ptr(out), dt, N, H * W, C, _stream())
        return out
    _need_gpu(x, scale, add_full, add_vec)
    x = x.contiguous()
    N, H, W, C = x.shape
    out = torch.empty_like(x)
    if add_full is not None:
        add_full = add_full.contiguous()
    launch("scale_add", _lib.load().arseg_scale_add_fwd, _ptr(x), _ptr(scale.contiguous()), _ptr(add_full),
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
        launch("head", _lib.load().arseg_head16_fwd, _ptr(p_nhwc), _nhwc_ld(p_nhwc), dt, _ptr(wf), _ptr(bf), _ptr(out), N, H * W, C, n_cls,
                1 if log_softmax else 0, _stream())
        return out
    _need_gpu(p_nhwc, wf, bf)
    N, H, W, C = p_nhwc.shape
    n_cls = wf.shape[0]
    out = torch.empty((N, n_cls, H, W), dtype=torch.float32, device=p_nhwc.device)
    launch("head", _lib.load().arseg_head_fwd, _ptr(p_nhwc), _nhwc_ld(p_nhwc), _ptr(wf), _ptr(bf), _ptr(out), N, H * W, C, n_cls,
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
    launch("frame_to_nhwc4", _lib.load().arseg_frame_to_nhwc4_fwd, _ptr(img), _ptr(out), N, H, W, h, w, _stream())
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
    launch("frame_u8_to_nhwc4", _lib.load().arseg_frame_u8_to_nhwc4_fwd, _ptr(img_u8), _ptr(out), N, H, W, h, w, m3, s3, _stream())
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
    launch("merge_motion", lib.arseg_merge_motion_fwd, _ptr(flows), _ptr(out), _ptr(ws), nbytes, F1 - 1, frame_start, H, W, _stream())
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
    launch("argmax_confusion", _lib.load().arseg_argmax_confusion_fwd, _ptr(logits), _ptr(label), _ptr(pred), _ptr(hist if label is not None else None), N,
                                                 n_cls, h, w, H, W, ignore_label, 1 if align_corners else 0, _stream())
    return pred, hist
