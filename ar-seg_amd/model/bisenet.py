"""BiSeNetV1-18 -- mirror of the reference's ``model/bisenet.py`` (everything reachable from
``BiSeNetV1(.., 'resnet18')`` / ``BiSeNetV1WithFuse``) on libarseg_hip.so.

Same class names, constructor arguments, ``forward`` / ``forward_phase1`` / ``forward_phase2``
signatures and ``state_dict`` keys, including the aliased registrations ``feat_conv_out`` ==
``conv_out.conv`` and ``final_conv`` == ``conv_out.conv_out`` (bisenet.py:428-430, 490-492).
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import BatchNorm2d

from .. import _lib, ops
from ..packing import PackedConv, PackedHead
from ._common import HipModule
from .attention import MyAttention


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(HipModule):
    """bisenet.py:31-60."""

    def __init__(self, in_chan, out_chan, stride=1):
        super().__init__()
        self.conv1 = conv3x3(in_chan, out_chan, stride)
        self.bn1 = BatchNorm2d(out_chan)
        self.conv2 = conv3x3(out_chan, out_chan)
        self.bn2 = BatchNorm2d(out_chan)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if in_chan != out_chan or stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_chan, out_chan, kernel_size=1, stride=stride, bias=False),
                                            BatchNorm2d(out_chan))

    def _pack(self, device):
        pk = {"c1": PackedConv.from_modules(self.conv1, self.bn1, _lib.ACT_RELU, device=device),
              "c2": PackedConv.from_modules(self.conv2, self.bn2, _lib.ACT_RELU, device=device)}
        if self.downsample is not None:
            pk["ds"] = PackedConv.from_modules(self.downsample[0], self.downsample[1], _lib.ACT_NONE, device=device)
        return pk

    def forward_nhwc(self, x):
        pk = self.packed()
        y = ops.conv2d(x, pk["c1"])
        sc = ops.conv2d(x, pk["ds"]) if "ds" in pk else x
        return ops.conv2d(y, pk["c2"], residual=sc)


def create_layer_basic(in_chan, out_chan, bnum, stride=1):
    layers = [BasicBlock(in_chan, out_chan, stride=stride)]
    for _ in range(bnum - 1):
        layers.append(BasicBlock(out_chan, out_chan, stride=1))
    return nn.Sequential(*layers)


class Resnet18(HipModule):
    """bisenet.py:70-94.  (The reference downloads ImageNet weights in ``init_weight``; offline, weights
    arrive through ``load_state_dict``.)"""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = create_layer_basic(64, 64, bnum=2, stride=1)
        self.layer2 = create_layer_basic(64, 128, bnum=2, stride=2)
        self.layer3 = create_layer_basic(128, 256, bnum=2, stride=2)
        self.layer4 = create_layer_basic(256, 512, bnum=2, stride=2)

    def _pack(self, device):
        return PackedConv.from_modules(self.conv1, self.bn1, _lib.ACT_RELU, device=device)

    def forward_nhwc(self, x4):
        x = ops.maxpool3x3s2(ops.conv2d(x4, self.packed()))
        for blk in self.layer1:
            x = blk.forward_nhwc(x)
        feats = []
        for layer in (self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x = blk.forward_nhwc(x)
            feats.append(x)
        return feats[0], feats[1], feats[2]


class ConvBNReLU(HipModule):
    """bisenet.py:162-186."""

    def __init__(self, in_chan, out_chan, ks=3, stride=1, padding=1, *args, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_chan, out_chan, kernel_size=ks, stride=stride, padding=padding, bias=False)
        self.bn = BatchNorm2d(out_chan)
        self.relu = nn.ReLU(inplace=True)

    def _pack(self, device):
        return PackedConv.from_modules(self.conv, self.bn, _lib.ACT_RELU, device=device)

    def forward_nhwc(self, x, out=None):
        return ops.conv2d(x, self.packed(), out=out)


class BiSeNetOutput(HipModule):
    """bisenet.py:207-223."""

    def __init__(self, in_chan, mid_chan, n_classes, up_factor=32, *args, **kwargs):
        super().__init__()
        self.up_factor = up_factor
        self.conv = ConvBNReLU(in_chan, mid_chan, ks=3, stride=1, padding=1)
        self.conv_out = nn.Conv2d(mid_chan, n_classes, kernel_size=1, bias=True)
        self.up = nn.Upsample(scale_factor=up_factor, mode='bilinear', align_corners=False)

    def _pack(self, device):
        return PackedHead(self.conv_out, device)

    def head_nhwc(self, mid):
        """conv_out (1x1 + bias) -> x up_factor bilinear (align_corners=False): NHWC feature -> NCHW logits."""
        hd = self.packed()
        lo = ops.head(mid, hd.wf, hd.bf, log_softmax=False)
        N, n_cls, h, w = lo.shape
        return ops.resize_nchw(lo, h * self.up_factor, w * self.up_factor, _lib.BILINEAR, False)

    def forward_nhwc(self, x):
        return self.head_nhwc(self.conv.forward_nhwc(x))


class AttentionRefinementModule(HipModule):
    """bisenet.py:226-260."""

    def __init__(self, in_chan, out_chan, *args, **kwargs):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, out_chan, ks=3, stride=1, padding=1)
        self.conv_atten = nn.Conv2d(out_chan, out_chan, kernel_size=1, bias=False)
        self.bn_atten = BatchNorm2d(out_chan)

    def _pack(self, device):
        return PackedConv.from_modules(self.conv_atten, self.bn_atten, _lib.ACT_SIGMOID, device=device)

    def forward_nhwc(self, x, add_full=None, add_vec=None):
        """feat * sigmoid(bn(conv(mean(feat)))) with the caller's following add fused in."""
        feat = self.conv.forward_nhwc(x)
        atten = ops.conv2d(ops.global_reduce(feat, _lib.REDUCE_MEAN), self.packed())
        return ops.scale_add(feat, atten, add_full=add_full, add_vec=add_vec)


class ContextPath(HipModule):
    """bisenet.py:263-306."""

    def __init__(self, backend, *args, **kwargs):
        super().__init__()
        if backend == 'resnet18':
            self.resnet = Resnet18()
        else:
            raise NotImplementedError("only the resnet18 backend is on the hot path (evaluation.py:24-36)")
        self.arm16 = AttentionRefinementModule(256, 128)
        self.arm32 = AttentionRefinementModule(512, 128)
        self.conv_head32 = ConvBNReLU(128, 128, ks=3, stride=1, padding=1)
        self.conv_head16 = ConvBNReLU(128, 128, ks=3, stride=1, padding=1)
        self.conv_avg = ConvBNReLU(512, 128, ks=1, stride=1, padding=0)
        self.up32 = nn.Upsample(scale_factor=2.)
        self.up16 = nn.Upsample(scale_factor=2.)

    def _pack(self, device):
        return {}

    def forward_nhwc(self, x4, out16=None):
        feat8, feat16, feat32 = self.resnet.forward_nhwc(x4)
        avg = self.conv_avg.forward_nhwc(ops.global_reduce(feat32, _lib.REDUCE_MEAN))
        feat32_sum = self.arm32.forward_nhwc(feat32, add_vec=avg)
        h32, w32 = feat32_sum.shape[1:3]
        feat32_up = ops.resize_nhwc(feat32_sum, 2 * h32, 2 * w32, _lib.NEAREST, False)          # nn.Upsample(scale_factor=2.)
        h16, w16 = feat16.shape[1:3]
        if (2 * h32, 2 * w32) != (h16, w16):                                                     # bisenet.py:298 (identity otherwise)
            feat32_up = ops.resize_nhwc(feat32_up, h16, w16, _lib.BILINEAR, True)
        feat32_up = self.conv_head32.forward_nhwc(feat32_up)
        feat16_sum = self.arm16.forward_nhwc(feat16, add_full=feat32_up)
        feat16_up = ops.resize_nhwc(feat16_sum, 2 * h16, 2 * w16, _lib.NEAREST, False)
        feat16_up = self.conv_head16.forward_nhwc(feat16_up, out=out16)
        return feat16_up, feat32_up


class SpatialPath(HipModule):
    """bisenet.py:326-340."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.conv1 = ConvBNReLU(3, 64, ks=7, stride=2, padding=3)
        self.conv2 = ConvBNReLU(64, 64, ks=3, stride=2, padding=1)
        self.conv3 = ConvBNReLU(64, 64, ks=3, stride=2, padding=1)
        self.conv_out = ConvBNReLU(64, 128, ks=1, stride=1, padding=0)

    def _pack(self, device):
        return {}

    def forward_nhwc(self, x4, out=None):
        f = self.conv3.forward_nhwc(self.conv2.forward_nhwc(self.conv1.forward_nhwc(x4)))
        return self.conv_out.forward_nhwc(f, out=out)


class FeatureFusionModule(HipModule):
    """bisenet.py:360-399."""

    def __init__(self, in_chan, out_chan, *args, **kwargs):
        super().__init__()
        self.convblk = ConvBNReLU(in_chan, out_chan, ks=1, stride=1, padding=0)
        self.conv = nn.Conv2d(out_chan, out_chan, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn = nn.BatchNorm2d(out_chan)

    def _pack(self, device):
        return PackedConv.from_modules(self.conv, self.bn, _lib.ACT_SIGMOID, device=device)

    def forward_nhwc(self, fcat):
        feat = self.convblk.forward_nhwc(fcat)
        atten = ops.conv2d(ops.global_reduce(feat, _lib.REDUCE_MEAN), self.packed())
        return ops.scale_add(feat, atten, add_full=feat)                                        # feat*atten + feat


class _BiSeBase(HipModule):
    SUPPORTS_16BIT = True

    def _build(self, n_classes, backend, aux_mode):
        self.cp = ContextPath(backend=backend)
        self.sp = SpatialPath()
        self.ffm = FeatureFusionModule(256, 256)
        self.conv_out = BiSeNetOutput(256, 256, n_classes, up_factor=8)
        self.feat_conv_out = self.conv_out.conv
        self.final_conv = self.conv_out.conv_out
        self.out_upsample = self.conv_out.up
        self.aux_mode = aux_mode
        if self.aux_mode == 'train':
            self.conv_out16 = BiSeNetOutput(128, 64, n_classes, up_factor=8)
            self.conv_out32 = BiSeNetOutput(128, 64, n_classes, up_factor=16)

    def _pack(self, device):
        return {}

    def _trunk_nhwc(self, x):
        """NCHW frame -> (feat_cp8, feat_cp16, middle_feat), all NHWC."""
        N, C, H, W = x.shape
        return self._trunk_nhwc4(ops.frame_ingest(x, H, W, self.storage_dtype))

    def phase1_nhwc4(self, x4, aux=True):
        """forward_phase1 on an NHWC4 frame: same outputs as the reference (aux heads included in 'train'
        aux_mode, bisenet.py:557-559) with the CReFF input left in NHWC.  ``aux=False`` (the build's fast paths, which read
        ``[-1]`` only as evaluation.py:190-191 does): the two training-only aux heads and their x8 / x16 upsamples are not evaluated."""
        cp8, cp16, mid = self._trunk_nhwc4(x4)
        if not aux:
            return (mid,)
        if self.aux_mode == 'train':
            return self.conv_out16.forward_nhwc(cp8), self.conv_out32.forward_nhwc(cp16), mid
        if self.aux_mode == 'eval':
            return (mid,)
        raise NotImplementedError

    def _trunk_nhwc4(self, x4):
        N, H, W, _ = x4.shape
        h8 = _half(_half(_half(H)))
        w8 = _half(_half(_half(W)))
        h16, w16 = _half(h8), _half(w8)
        ch, cw = 2 * h16, 2 * w16                                  # context-path 1/8 size = 2 x feat16 size
        fcat = torch.empty((N, ch, cw, 256), dtype=x4.dtype, device=x4.device)          # cat([fsp, fcp], 1) in place
        feat_cp8, feat_cp16 = self.cp.forward_nhwc(x4, out16=fcat[..., 128:])
        if (h8, w8) == (ch, cw):
            self.sp.forward_nhwc(x4, out=fcat[..., :128])
        else:                                                                                  # bisenet.py:442
            ops.resize_nhwc(self.sp.forward_nhwc(x4), ch, cw, _lib.BILINEAR, True, out=fcat[..., :128])
        feat_fuse = self.ffm.forward_nhwc(fcat)
        return feat_cp8, feat_cp16, self.feat_conv_out.forward_nhwc(feat_fuse)

    def forward_keyframe(self, x):
        """The keyframe's pass as the video pipeline needs it (evaluation.py:173-174 reads ``[-1]``, the frame's own segmentation is ``[0]``):
        -> (logits NCHW at frame resolution, middle feature NHWC).  Same arithmetic as ``forward``; the training-only aux heads
        (bisenet.py:455-457) and their x8 / x16 upsamples are not evaluated."""
        self._check_inference()
        _, _, mid = self._trunk_nhwc(x)
        return self.conv_out.head_nhwc(mid), mid

    def _forward_normal(self, x):
        self._check_inference()
        cp8, cp16, mid = self._trunk_nhwc(x)
        feat_out = self.conv_out.head_nhwc(mid)
        if self.aux_mode == 'train':
            return feat_out, self.conv_out16.forward_nhwc(cp8), self.conv_out32.forward_nhwc(cp16), ops.as_nchw(mid)
        if self.aux_mode == 'eval':
            return feat_out,
        if self.aux_mode == 'pred':
            pred, _ = ops.argmax_confusion(feat_out, None, feat_out.shape[2], feat_out.shape[3])
            return pred.long()
        raise NotImplementedError


def _half(n):
    return (n - 1) // 2 + 1


class BiSeNetV1(_BiSeBase):
    """bisenet.py:419-477 (the HR / keyframe branch)."""

    def __init__(self, n_classes, backend, aux_mode='train', *args, **kwargs):
        super().__init__()
        self._build(n_classes, backend, aux_mode)

    def forward(self, x):
        return self._forward_normal(x)


class BiSeNetV1WithFuse(_BiSeBase):
    """bisenet.py:481-596 (the LR branch with CReFF on the 256-channel 1/8-resolution feature)."""

    def __init__(self, n_classes, backend, aux_mode='train', attention_type='local', atten_k=7, *args, **kwargs):
        super().__init__()
        self._build(n_classes, backend, aux_mode)
        self.middle_dim = 256
        if attention_type != 'local':
            raise NotImplementedError("only attention_type='local' is on the hot path")
        self.fuse_attention = MyAttention(self.middle_dim, kH=atten_k, kW=atten_k)

    def forward(self, x, mode='normal', ref_p=None):
        if mode == 'normal':
            return self._forward_normal(x)
        if mode == 'merge':
            if self.aux_mode == 'train':
                feat_out16, feat_out32, middle_feat = self.forward_phase1(x)
            elif self.aux_mode == 'eval':
                middle_feat = self.forward_phase1(x)
            else:
                raise NotImplementedError
            out, out_p = self.forward_phase2(middle_feat, ref_p)
            if self.aux_mode == 'train':
                return out, feat_out16, feat_out32, out_p
            return out,
        raise ValueError(mode)

    def forward_phase1(self, x):
        self._check_inference()
        cp8, cp16, mid = self._trunk_nhwc(x)
        if self.aux_mode == 'train':
            return self.conv_out16.forward_nhwc(cp8), self.conv_out32.forward_nhwc(cp16), ops.as_nchw(mid)
        if self.aux_mode == 'eval':
            return ops.as_nchw(mid)
        raise NotImplementedError

    def phase2_c8(self, mid_nhwc, ref_c8):
        """Kernel-layout phase 2: -> (logits NCHW at full resolution, p C8)."""
        hd = self.conv_out.packed()
        p_c8, lo = self.fuse_attention.fuse_c8(ref_c8, mid_nhwc, head=(hd.wf, hd.bf), log_softmax=False)
        N, n_cls, h, w = lo.shape
        return ops.resize_nchw(lo, 8 * h, 8 * w, _lib.BILINEAR, False), p_c8                   # out_upsample

    def phase2_warp(self, mid_nhwc, refs_nhwc, mv_q, upsample=True):
        """Phase 2 with the MV warp in front (fast path): -> (logits NCHW at full resolution, p C8).  ``upsample=False`` returns the
        head's logits at 1/8 resolution instead, for a caller that fuses out_upsample (x8 bilinear, align_corners=False,
        bisenet.py:215-216) into the argmax (ops.argmax_confusion(..., align_corners=False)): 159 MB per frame never written."""
        hd = self.conv_out.packed()
        p_c8, lo = self.fuse_attention.fuse_warp(refs_nhwc, mv_q, mid_nhwc, head=(hd.wf, hd.bf), log_softmax=False)
        if not upsample:
            return lo, p_c8
        N, n_cls, h, w = lo.shape
        return ops.resize_nchw(lo, 8 * h, 8 * w, _lib.BILINEAR, False), p_c8                   # out_upsample

    def forward_phase2(self, middle_feat, ref_p):
        self._check_inference()
        if ops.is16(middle_feat) or ops.is16(ref_p):      # 16-bit storage through the NCHW interface: the CReFF stage itself is fp32
            middle_feat, ref_p = ops.cast(middle_feat, torch.float32), ops.cast(ref_p, torch.float32)
        ref_c8 = ops.to_c8(ops.to_nhwc(ref_p), _lib.NHWC) if ops.is_nhwc_view(ref_p) else ops.to_c8(ref_p, _lib.NCHW)
        out, p_c8 = self.phase2_c8(ops.to_nhwc(middle_feat), ref_c8)
        return out, ops.as_nchw(ops.from_c8(p_c8, _lib.NHWC))
