"""CamVid PSPNet-18 -- mirror of the reference's ``model/pspnet.py:14-231`` on libarseg_hip.so.

Same class names, constructor keywords, ``forward`` / ``forward_phase1`` / ``forward_phase2``
signatures and ``state_dict`` keys.  Tensors at the interface are logical NCHW; outputs produced
here are physically NHWC (``torch.channels_last`` strides), which every entry point also accepts,
so ``ref_p`` flows HR net -> warpFeature -> forward_phase2 without layout copies.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib, ops
from ..packing import PackedConv, PackedHead
from . import extractors
from ._common import HipModule
from .attention import MyAttention


class PSPModule(HipModule):
    """model/pspnet.py:14-31."""

    def __init__(self, features, out_features=1024, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.stages = nn.ModuleList([self._make_stage(features, size) for size in sizes])
        self.bottleneck = nn.Conv2d(features * (len(sizes) + 1), out_features, kernel_size=1)
        self.relu = nn.ReLU()
        self.sizes = tuple(sizes)
        self.features = features

    def _make_stage(self, features, size):
        prior = nn.AdaptiveAvgPool2d(output_size=(size, size))
        conv = nn.Conv2d(features, features, kernel_size=1, bias=False)
        return nn.Sequential(prior, conv)

    def _pack(self, device):
        """The pyramid is folded at pack time.  Every op between the pooled maps and the bottleneck output is linear
        (1x1 stage conv, bilinear upsample, the stage's slice of the 1x1 bottleneck conv) and 1x1 convs commute with
        upsampling, so
            bottleneck(cat(up(conv_s(pool_s f)), f)) = W_f f + sum_s up((W_s . Wconv_s) pool_s f) + b
        with W_s the bottleneck columns of level s.  The 2560-channel concat and 80% of the bottleneck MACs disappear;
        fp32 rounding differs from the reference's order of summation at the 1e-6 level."""
        C, n = self.features, len(self.sizes)
        wb = self.bottleneck.weight.detach().double().cpu().reshape(self.bottleneck.out_channels, -1)      # [1024, 2560]
        comb = [wb[:, i * C:(i + 1) * C] @ self.stages[i][1].weight.detach().double().cpu().reshape(C, C) for i in range(n)]
        w_prior = torch.cat(comb, dim=1).float()                                                            # [1024, n*C]
        w_feat = wb[:, n * C:].float()
        return {"prior": PackedConv(w_prior, None, None, act=_lib.ACT_NONE, device=device),
                "feat": PackedConv(w_feat, self.bottleneck.bias, None, act=_lib.ACT_RELU, device=device)}

    def forward_nhwc(self, feats):
        pk = self.packed()
        N, h, w, C = feats.shape
        rows = sum(s * s for s in self.sizes)
        # block-structured matrix: the rows of level i hold its pooled map in columns [i*C, (i+1)*C), zeros elsewhere (written by the
        # pooling launches themselves: a torch.zeros here was a fill launch per forward, 2.4 % of the traced GPU time in round 2)
        pooled = ops.psp_pool_matrix(feats, self.sizes)
        t = ops.conv2d(pooled, pk["prior"])                                   # [N, rows, 1, 1024]: all levels, one launch
        if ops.gemm_x3_enabled() and rows <= 64 and C % 32 == 0 and not ops.is16(feats) and ops.psp_x3_foldable(pk["feat"]):
            # the pyramid sum as two more K steps of the bottleneck GEMM (interpolation matrix x per-image pyramid terms), result in split rows
            return ops.psp_bottleneck_x3(feats, t.reshape(N, rows, -1), pk["feat"], self.sizes)
        prior = ops.psp_prior_sum(t.reshape(N, rows, -1), self.sizes, h, w)   # sum_s upsample(t_s), F.upsample default mode
        # + W_f f + b, ReLU.  The only consumer is up_1's low-resolution tap GEMM: with the LDS-DMA GEMM enabled the result is written as
        # split rows (ops.SplitRows) and up_1 stages it without a conversion; otherwise this is an ordinary fp32 NHWC tensor
        return ops.conv2d(feats, pk["feat"], residual=prior, out_split=True)


class PSPUpsample(HipModule):
    """model/pspnet.py:34-46."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, padding=1), nn.BatchNorm2d(out_channels), nn.PReLU())

    def _pack(self, device):
        return PackedConv.from_modules(self.conv[0], self.conv[1], _lib.ACT_PRELU, float(self.conv[2].weight.item()), device=device)

    def forward_nhwc(self, x, out_split=False):
        # x2 bilinear upsample (F.upsample default) + conv + BN + PReLU.  out_split: the only consumer is another PSPUpsample -- on the split-row
        # route (x is an ops.SplitRows) the result is written as split rows too; otherwise the flag is ignored
        return ops.conv2d(x, self.packed(), up2=True, out_split=out_split)


class _PSPBase(HipModule):
    """Everything PSPNet and PSPNetWithFuse share (the reference duplicates the code)."""

    def _build(self, input_channel, n_classes, sizes, psp_size, deep_features_size, backend, pretrained):
        self.feats = getattr(extractors, backend)(pretrained, input_channel=input_channel)
        self.psp = PSPModule(psp_size, 1024, sizes)
        self.drop_1 = nn.Dropout2d(p=0.3)
        self.up_1 = PSPUpsample(1024, 256)
        self.up_2 = PSPUpsample(256, 64)
        self.up_3 = PSPUpsample(64, 64)
        self.drop_2 = nn.Dropout2d(p=0.15)
        self.final_conv = nn.Conv2d(64, n_classes, kernel_size=1)
        self.final_logsoftmax = nn.LogSoftmax()
        self.classifier = nn.Sequential(nn.Linear(deep_features_size, 256), nn.ReLU(), nn.Linear(256, n_classes))

    def _pack(self, device):
        return {"head": PackedHead(self.final_conv, device),
                "cls0": PackedConv.from_modules(self.classifier[0], None, _lib.ACT_RELU, device=device),
                "cls2": PackedConv.from_modules(self.classifier[2], None, _lib.ACT_NONE, device=device)}

    def _trunk_nhwc(self, x):
        """NCHW frame -> (aux logits [N,n_cls], p NHWC [N,H',W',64])."""
        N, C, H, W = x.shape
        return self.phase1_nhwc4(ops.frame_to_nhwc4(x, H, W))

    def phase1_nhwc4(self, x4, aux=True):
        """The backbone on an NHWC4 frame (pspnet.py:198-217): -> (aux logits [N,n_cls], p NHWC).  ``aux=False`` (the build's fast paths, which
        read ``[-1]`` only as evaluation.py:190-191 does): the training-only auxiliary classifier (pspnet.py:92-94) is not evaluated -> (None, p)."""
        N, H, W, _ = x4.shape
        f, class_f = self.feats.forward_nhwc(x4)
        p = self.psp.forward_nhwc(f)            # drop_1 / drop_2: identity in eval
        p = self.up_1.forward_nhwc(p, out_split=True)
        p = self.up_2.forward_nhwc(p)
        p = self.up_3.forward_nhwc(p)
        if not aux:
            return None, p
        pk = self.packed()
        aux = ops.global_reduce(class_f, _lib.REDUCE_MAX)
        aux = ops.conv2d(ops.conv2d(aux, pk["cls0"]), pk["cls2"])
        return aux.reshape(N, -1), p

    def _final(self, p_nhwc, H, W):
        """final_conv -> interpolate(align_corners=True) to (H,W) -> LogSoftmax (pspnet.py:96-98).  The 1x1 conv and
        the bilinear resize are both linear and commute, so an (uncommon) size mismatch is handled by resizing p."""
        if p_nhwc.shape[1] != H or p_nhwc.shape[2] != W:
            p_nhwc = ops.resize_nhwc(p_nhwc, H, W, _lib.BILINEAR, True)
        hd = self.packed()["head"]
        return ops.head(p_nhwc, hd.wf, hd.bf, log_softmax=True)

    def _forward_normal(self, x):
        self._check_inference()
        N, C, H, W = x.shape
        aux, p = self._trunk_nhwc(x)
        return self._final(p, H, W), aux, ops.as_nchw(p)

    def forward_keyframe(self, x):
        """The keyframe's pass as the video pipeline needs it (evaluation.py:173-174 reads ``[-1]``, the frame's own segmentation is ``[0]``):
        -> (log-probs NCHW, p NHWC).  Same arithmetic as ``forward``; the training-only auxiliary classifier output is not evaluated."""
        self._check_inference()
        N, C, H, W = x.shape
        _, p = self.phase1_nhwc4(ops.frame_to_nhwc4(x, H, W), aux=False)
        return self._final(p, H, W), p


class PSPNet(_PSPBase):
    """model/pspnet.py:49-100 (the HR / keyframe branch)."""

    def __init__(self, input_channel=3, n_classes=18, sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024,
                 backend='resnet34', pretrained=True):
        super().__init__()
        self._build(input_channel, n_classes, sizes, psp_size, deep_features_size, backend, pretrained)

    def forward(self, x):
        return self._forward_normal(x)


class PSPNetWithFuse(_PSPBase):
    """model/pspnet.py:103-231 (the LR branch with CReFF on the 64-channel full-resolution feature)."""

    def __init__(self, input_channel=3, n_classes=18, sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024,
                 backend='resnet34', pretrained=True, attention_type='local', atten_k=7):
        super().__init__()
        self._build(input_channel, n_classes, sizes, psp_size, deep_features_size, backend, pretrained)
        self.middle_dim = 64
        self.attention_type = attention_type
        if attention_type != 'local':
            raise NotImplementedError("only attention_type='local' (MyAttention) is on the hot path; the reference's "
                                      "evaluation never constructs the ablation variants")
        self.fuse_attention = MyAttention(self.middle_dim, kH=atten_k, kW=atten_k)

    def forward(self, x, mode='normal', ref_p=None):
        if mode == 'normal':
            return self._forward_normal(x)
        if mode == 'merge':
            out_cls, out_p = self.forward_phase1(x)
            out, out_p = self.forward_phase2(out_p, ref_p)
            return out, out_cls, out_p
        raise ValueError(mode)

    def forward_phase1(self, x):
        self._check_inference()
        aux, p = self._trunk_nhwc(x)
        return aux, ops.as_nchw(p)

    def phase2_c8(self, p_nhwc, ref_c8, want_p=True):
        """Kernel-layout phase 2 (fast path): LR feature NHWC + warped HR feature C8 -> (log-probs NCHW, p C8)."""
        hd = self.packed()["head"]
        p_c8, out = self.fuse_attention.fuse_c8(ref_c8, p_nhwc, head=(hd.wf, hd.bf), log_softmax=True)
        return out, p_c8

    def phase2_warp(self, p_nhwc, refs_nhwc, mv_q):
        """Phase 2 with the MV warp fused in (fast path): LR feature NHWC, un-warped keyframe features NHWC [Hp,Wp,C] (one per
        frame), int16 MVs -> (log-probs NCHW, p C8)."""
        hd = self.packed()["head"]
        p_c8, out = self.fuse_attention.fuse_warp(refs_nhwc, mv_q, p_nhwc, head=(hd.wf, hd.bf), log_softmax=True)
        return out, p_c8

    def forward_phase2(self, p, ref_p):
        self._check_inference()
        N, C, H, W = ref_p.shape
        ref_c8 = ops.to_c8(ops.to_nhwc(ref_p), _lib.NHWC) if ops.is_nhwc_view(ref_p) else ops.to_c8(ref_p, _lib.NCHW)
        out, p_c8 = self.phase2_c8(ops.to_nhwc(p), ref_c8)
        # F.interpolate(out, (H,W), align_corners=True) (pspnet.py:227) is the identity: out already is H x W
        return out, ops.as_nchw(ops.from_c8(p_c8, _lib.NHWC))
