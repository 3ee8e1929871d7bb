"""CReFF fusion module -- mirror of the reference's ``model/attention.py`` (hot-path part only).

``MyAttention`` (model/attention.py:157-229) is the only variant the reference's evaluation ever
constructs (model/pspnet.py:135-136, model/bisenet.py:500-501); the 17 ablation variants are out of
scope.  ``f_similar`` / ``f_weighting`` (model/attention.py:13-53) keep their names and call the
stand-alone HIP kernels; their backward passes (training) are not part of this path and raise.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function

from .. import _lib, ops, packing
from ._common import HipModule


class similarFunction(Function):
    """model/attention.py:13-30 (forward only)."""

    @staticmethod
    def forward(ctx, x_ori, x_loc, kH, kW):
        return ops.local_similar(x_ori, x_loc, kH, kW)

    @staticmethod
    def backward(ctx, grad_outputs):
        raise NotImplementedError("localAttention backward (training) is outside the LR-branch inference hot path")


class weightingFunction(Function):
    """model/attention.py:33-50 (forward only)."""

    @staticmethod
    def forward(ctx, x_ori, x_weight, kH, kW):
        return ops.local_weighting(x_ori, x_weight, kH, kW)

    @staticmethod
    def backward(ctx, grad_outputs):
        raise NotImplementedError("localAttention backward (training) is outside the LR-branch inference hot path")


f_similar = similarFunction.apply
f_weighting = weightingFunction.apply


class MyAttention(HipModule):
    """model/attention.py:157-229.  Note the reference's argument order ``(feat_dim, kW, kH)``."""

    def __init__(self, feat_dim, kW, kH):
        super().__init__()
        self.lr_query_conv = nn.Conv2d(feat_dim, feat_dim, kernel_size=3, padding=1, groups=feat_dim)
        self.hr_key_conv = nn.Conv2d(feat_dim, feat_dim, kernel_size=3, padding=1, groups=feat_dim)
        self.hr_value_conv = nn.Conv2d(feat_dim, feat_dim, kernel_size=3, padding=1, groups=feat_dim)
        self.softmax = nn.Softmax(dim=3)
        self.kW = kW
        self.kH = kH
        # (no init_weight: inference only, parameters always come from a state_dict)

    def _pack(self, device):
        return packing.PackedAttention(self, device)

    def fuse_c8(self, hr_c8, lr_nhwc, head=None, log_softmax=False):
        """Kernel-layout entry: warped HR feature in C8, LR feature NHWC -> (p C8, logits NCHW | None)."""
        return ops.creff(hr_c8, lr_nhwc, self.packed(), head, log_softmax, self.kH, self.kW)

    def fuse_warp(self, refs_nhwc, mv_q, lr_nhwc, head=None, log_softmax=False):
        """warpFeature + forward (+ head) on the kernel layouts: un-warped keyframe features NHWC [Hp,Wp,C] (one per frame),
        int16 quarter-pel MVs [B,H,W,2], LR feature NHWC -> (p C8, logits NCHW | None).  One kernel for C == 64
        (arseg_creff_warp_fwd); arseg_warp_mvq_fwd + arseg_creff_fwd otherwise."""
        return ops.creff_warp(refs_nhwc, mv_q, lr_nhwc, self.packed(), head, log_softmax, self.kH, self.kW)

    def forward(self, hr_feat, lr_feat):
        """hr_feat [N,C,H,W], lr_feat [N,C,h,w] (logical NCHW, any strides) -> [N,C,H,W]."""
        hr_c8 = ops.to_c8(ops.to_nhwc(hr_feat), _lib.NHWC) if ops.is_nhwc_view(hr_feat) else ops.to_c8(hr_feat, _lib.NCHW)
        p_c8, _ = self.fuse_c8(hr_c8, ops.to_nhwc(lr_feat))
        return ops.as_nchw(ops.from_c8(p_c8, _lib.NHWC))
