"""Inputs of the hot path as the reference's datasets hold them on disk.

* motion vectors: ``.bin`` files of int16 quarter-pel ``[H,W,2]`` (dataset/camvid.py:624-626, dataset/cityscapes.py:282-285).
  The reference reads them with ``np.fromfile(..., np.short).reshape(H,W,2) / 4`` into float64 pixels and ships 16 B/pixel to
  the GPU; here the int16 array is uploaded as is (4 B/pixel) and consumed by ``ops.warp_mvq`` (MV resize + warp fused).
* decoded frames: uint8 HWC; ``ToTensor`` + ``Normalize`` (dataset/camvid.py:503-506) and the evaluator's downscale
  (evaluation.py:186-188) run in one kernel, ``ops.frame_u8_to_nhwc4``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

CAMVID_MEAN, CAMVID_STD = (0.39068785, 0.40521392, 0.41434407), (0.29652068, 0.30514979, 0.30080369)      # camvid.py:505
CITY_BISE_MEAN, CITY_BISE_STD = (0.3257, 0.3690, 0.3223), (0.2112, 0.2148, 0.2115)                          # cityscapes.py:211-212


def read_mv_bin(path, H: int, W: int) -> np.ndarray:
    """int16 quarter-pel motion vectors [H,W,2] (x, y) of one non-keyframe, accumulated back to its keyframe."""
    mv = np.fromfile(path, dtype=np.int16)
    if mv.size != H * W * 2:
        raise ValueError(f"{path}: expected {H * W * 2} int16 values for a {H}x{W} frame, found {mv.size}")
    return mv.reshape(H, W, 2)


def mv_to_device(mv_q: np.ndarray, device) -> torch.Tensor:
    """[H,W,2] or [N,H,W,2] int16 -> device tensor [N,H,W,2] (the layout ops.warp_mvq takes)."""
    t = torch.from_numpy(np.ascontiguousarray(mv_q, dtype=np.int16))
    return (t.unsqueeze(0) if t.dim() == 3 else t).to(device)


def frames_to_nhwc4(frames_u8, h: int, w: int, mean=CAMVID_MEAN, std=CAMVID_STD, device="cuda") -> torch.Tensor:
    """uint8 frames [H,W,3] / [N,H,W,3] (numpy or tensor) -> normalised NHWC4 [N,h,w,4] on the GPU."""
    t = torch.as_tensor(np.ascontiguousarray(frames_u8)) if not torch.is_tensor(frames_u8) else frames_u8
    if t.dim() == 3:
        t = t.unsqueeze(0)
    return ops.frame_u8_to_nhwc4(t.to(device), h, w, mean, std)
