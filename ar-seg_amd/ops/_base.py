"""Plumbing shared by the op wrappers: stream / pointer marshalling, argument checks, the caller-owned workspace, NHWC strides and
the sticky operand-range word of the split-fp16 back end."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _lib
from ._config import sw


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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


# Operand-range safety of the split-fp16 back end (include/arseg_hip.h, ARSEG_MATH_F16X3: the hi/lo pair carries 22 bits up to |x| = 65504
# and clamps beyond 131008; the Winograd route multiplies TRANSFORMED activations, ~10x the input).
#   * default ("device"): every f16x3 conv gets a sticky device word (arseg_conv_desc.range_flag); the kernels set it when an activation
#     they multiply exceeds 65504 -- no host synchronisation, capturable.  The host reads the word once per batch of launches with
#     ops.range_tripped() (one sync) and repeats the batch under ops.set_conv_math("f32") -- evaluation.Eval*Res do, bench.py reports it.
#   * ARSEG_CONV_RANGE_GUARD=host: the round-2 validation mode -- amax of every conv input with one host sync per conv, the layer is
#     evaluated with the fp32 MFMA back end at once (not capturable).   * ARSEG_CONV_RANGE_GUARD=0: off.
RANGE_LIMIT = 2.0e4           # host-mode guard: inputs beyond this go to the fp32 back end
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
    if sw.RANGE_MODE == "device" and d.math == _lib.MATH_F16X3:
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
