"""Cityscapes PSPNet-18 -- mirror of the reference's ``model/pspnet_semseg.py:12-250`` on libarseg_hip.so
(what ``evaluation.py:27,34`` builds for ``cityscapes-psp18``).

Same class names, constructor keywords, ``forward`` / ``forward_phase1`` / ``forward_phase2`` signatures and
``state_dict`` keys (``layer0.0`` ... ``cls.4``, the ``final_conv`` alias of ``cls.4``, ``aux.*``, ``ppm.features.i.{1,2}``,
``fuse_attention.*``).  Differences from the CamVid PSPNet (model/pspnet.py): the pyramid stages carry BatchNorm + ReLU and
are upsampled with ``align_corners=True`` (no linear folding possible), the feature ``p`` is the 512-channel 1/8-resolution
output of ``cls[:-1]``, CReFF therefore runs at C=512 on the small map (matrix-core kernel), and phase 2 returns raw logits at
1/8 resolution.  Only ``layers=18`` is on the hot path.
"""
from __future__ import annotations

from torch import nn

from .. import _lib, ops
from ..packing import PackedConv, PackedHead
from . import extractors
from ._common import HipModule
from .attention import MyAttention


class PPM(HipModule):
    """model/pspnet_semseg.py:12-30."""

    def __init__(self, in_dim, reduction_dim, bins):
        super().__init__()
        self.features = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(b), nn.Conv2d(in_dim, reduction_dim, kernel_size=1, bias=False),
                                                     nn.BatchNorm2d(reduction_dim), nn.ReLU(inplace=True)) for b in bins])
        self.bins, self.in_dim, self.reduction_dim = tuple(bins), in_dim, reduction_dim

    def _pack(self, device):
        return [PackedConv.from_modules(f[1], f[2], _lib.ACT_RELU, device=device) for f in self.features]

    def forward_nhwc(self, cat):
        """``cat``: NHWC buffer [N,h,w,in_dim + len(bins)*reduction_dim] whose first in_dim channels hold the input; the
        pyramid levels are written into the remaining channel slices (torch.cat([x, level_1, ...], 1) without a copy)."""
        pk = self.packed()
        N, h, w, _ = cat.shape
        x = cat[..., :self.in_dim]
        for i, b in enumerate(self.bins):
            lvl = ops.conv2d(ops.adaptive_avgpool(x, b, b), pk[i])                          # 1x1 conv + BN + ReLU on the b x b map
            off = self.in_dim + i * self.reduction_dim
            ops.resize_nhwc(lvl, h, w, _lib.BILINEAR, True, out=cat[..., off:off + self.reduction_dim])
        return cat


class _SemsegBase(HipModule):
    """What PSPNet and PSPNetWithFuse share (the reference duplicates the code)."""

    def _build(self, layers, bins, dropout, classes, zoom_factor, feat_dim, use_ppm, pretrained):
        if layers != 18:
            raise NotImplementedError("only layers=18 (cityscapes-psp18, evaluation.py:27) is on the hot path")
        assert feat_dim % len(bins) == 0 and classes > 1 and zoom_factor in (1, 2, 4, 8)
        self.zoom_factor, self.use_ppm, self.feat_dim = zoom_factor, use_ppm, feat_dim
        resnet = extractors.resnet18(pretrained=pretrained)
        self.layer0 = nn.Sequential(resnet.conv1, resnet.bn1, resnet.relu, resnet.maxpool)
        self.layer1, self.layer2, self.layer3, self.layer4 = resnet.layer1, resnet.layer2, resnet.layer3, resnet.layer4
        # the reference re-dilates the second conv of every block of layer3 / layer4 (pspnet_semseg.py:59-68); the packer reads
        # stride / padding / dilation from these modules, so the same attribute edits apply here
        for layer, d in ((self.layer3, 2), (self.layer4, 4)):
            for n, m in layer.named_modules():
                if 'conv2' in n:
                    m.dilation, m.padding, m.stride = (d, d), (d, d), (1, 1)
                elif 'downsample.0' in n:
                    m.stride = (1, 1)
        fea_dim = feat_dim
        if use_ppm:
            self.ppm = PPM(fea_dim, int(fea_dim / len(bins)), bins)
            fea_dim *= 2
        self.cls = nn.Sequential(nn.Conv2d(fea_dim, 512, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(512), nn.ReLU(inplace=True),
                                 nn.Dropout2d(p=dropout), nn.Conv2d(512, classes, kernel_size=1))

    def _build_aux(self, feat_dim, dropout, classes):
        # created because a freshly constructed module is in training mode (pspnet_semseg.py:82,183): part of the state_dict
        self.aux = nn.Sequential(nn.Conv2d(feat_dim // 2, 256, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(256),
                                 nn.ReLU(inplace=True), nn.Dropout2d(p=dropout), nn.Conv2d(256, classes, kernel_size=1))

    def _pack(self, device):
        return {"stem": PackedConv.from_modules(self.layer0[0], self.layer0[1], _lib.ACT_RELU, device=device),
                "cls0": PackedConv.from_modules(self.cls[0], self.cls[1], _lib.ACT_RELU, device=device),
                "head": PackedHead(self.cls[4], device),
                "aux0": PackedConv.from_modules(self.aux[0], self.aux[1], _lib.ACT_RELU, device=device),
                "auxh": PackedHead(self.aux[4], device)}

    def phase1_nhwc4(self, x4, aux=True):
        """(``aux`` is accepted for symmetry with the other networks: this trunk evaluates no auxiliary output.)
        NHWC4 frame -> (x_tmp = layer3 output NHWC [N,h/8,w/8,256], p NHWC [N,h/8,w/8,512])  (pspnet_semseg.py:219-231)."""
        pk = self.packed()
        x = ops.maxpool3x3s2(ops.conv2d(x4, pk["stem"]))
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x = blk.forward_nhwc(x)
        x_tmp = x
        N, h, w, _ = x.shape
        C = self.feat_dim
        cat = x.new_empty((N, h, w, 2 * C if self.use_ppm else C))
        n4 = len(self.layer4)
        for i, blk in enumerate(self.layer4):                    # layer4's output lands in the first channels of the concat
            x = blk.forward_nhwc(x, out=cat[..., :C] if i == n4 - 1 else None)
        if self.use_ppm:
            self.ppm.forward_nhwc(cat)
        return x_tmp, ops.conv2d(cat, pk["cls0"])               # cls[:-1]: conv3x3 + BN + ReLU (Dropout2d = identity in eval)

    def _logits_up(self, feat_nhwc, head, H, W):
        """1x1 classifier -> F.interpolate(size=(H,W), bilinear, align_corners=True) unless zoom_factor == 1."""
        out = ops.head(feat_nhwc, head.wf, head.bf, log_softmax=False)
        if self.zoom_factor != 1:
            out = ops.resize_nchw(out, H, W, _lib.BILINEAR, True)
        return out

    def _aux(self, x_tmp, H, W):
        pk = self.packed()
        return self._logits_up(ops.conv2d(x_tmp, pk["aux0"]), pk["auxh"], H, W)


class PSPNet(_SemsegBase):
    """model/pspnet_semseg.py:33-115."""

    def __init__(self, layers=50, bins=(1, 2, 3, 6), dropout=0.1, classes=2, zoom_factor=8, feat_dim=2048, use_ppm=True,
                 criterion=None, pretrained=True):
        super().__init__()
        self.criterion = criterion
        self._build(layers, bins, dropout, classes, zoom_factor, feat_dim, use_ppm, pretrained)
        self._build_aux(feat_dim, dropout, classes)

    def forward(self, x, y=None):
        self._check_inference()
        N, C, H, W = x.shape
        _, p = self.phase1_nhwc4(ops.frame_to_nhwc4(x, H, W))
        return (self._logits_up(p, self.packed()["head"], H, W),)


class PSPNetWithFuse(_SemsegBase):
    """model/pspnet_semseg.py:117-250 (HR branch with mode='normal', LR branch through forward_phase1 / forward_phase2)."""

    def __init__(self, layers=50, bins=(1, 2, 3, 6), dropout=0.1, classes=2, zoom_factor=8, feat_dim=2048, use_ppm=True,
                 criterion=None, pretrained=True, attention_type='local', atten_k=7):
        super().__init__()
        self.criterion = criterion
        self._build(layers, bins, dropout, classes, zoom_factor, feat_dim, use_ppm, pretrained)
        self.final_conv = self.cls[-1]                          # alias: both key sets appear in the state_dict, as in the reference
        self._build_aux(feat_dim, dropout, classes)
        self.middle_dim = 512
        if attention_type != 'local':
            raise NotImplementedError("only attention_type='local' (MyAttention) is on the hot path")
        self.fuse_attention = MyAttention(self.middle_dim, kH=atten_k, kW=atten_k)

    def forward(self, x, mode='normal', ref_p=None):
        self._check_inference()
        N, C, H, W = x.shape
        x_tmp, p = self.phase1_nhwc4(ops.frame_to_nhwc4(x, H, W))
        if mode == 'normal':
            out = self._logits_up(p, self.packed()["head"], H, W)
            return out, self._aux(x_tmp, H, W), ops.as_nchw(p)
        if mode == 'merge':                                     # logits stay at 1/8 resolution in this mode (pspnet_semseg.py:202-206)
            out, p_c8 = self.phase2_c8(p, self._ref_c8(ref_p))
            return out, self._aux(x_tmp, H, W), ops.as_nchw(ops.from_c8(p_c8, _lib.NHWC))
        raise ValueError(mode)

    def forward_phase1(self, x):
        self._check_inference()
        N, C, H, W = x.shape
        x_tmp, p = self.phase1_nhwc4(ops.frame_to_nhwc4(x, H, W))
        return ops.as_nchw(x_tmp), ops.as_nchw(p)

    def forward_keyframe(self, x):
        """The keyframe's pass as the video pipeline needs it (evaluation.py:173-174 reads ``[-1]``): -> (logits NCHW, p NHWC); the
        training-only aux head (pspnet_semseg.py:196-199) is not evaluated."""
        self._check_inference()
        N, C, H, W = x.shape
        _, p = self.phase1_nhwc4(ops.frame_to_nhwc4(x, H, W))
        return self._logits_up(p, self.packed()["head"], H, W), p

    @staticmethod
    def _ref_c8(ref_p):
        return ops.to_c8(ops.to_nhwc(ref_p), _lib.NHWC) if ops.is_nhwc_view(ref_p) else ops.to_c8(ref_p, _lib.NCHW)

    def phase2_c8(self, p_nhwc, ref_c8):
        """Kernel-layout phase 2: LR feature NHWC + (warped) HR feature C8 -> (logits NCHW at feature resolution, p C8)."""
        hd = self.packed()["head"]
        p_c8, out = self.fuse_attention.fuse_c8(ref_c8, p_nhwc, head=(hd.wf, hd.bf), log_softmax=False)
        return out, p_c8

    def phase2_warp(self, p_nhwc, refs_nhwc, mv_q):
        """Phase 2 with the MV warp in front (fast path): -> (logits NCHW at feature resolution, p C8)."""
        hd = self.packed()["head"]
        p_c8, out = self.fuse_attention.fuse_warp(refs_nhwc, mv_q, p_nhwc, head=(hd.wf, hd.bf), log_softmax=False)
        return out, p_c8

    def forward_phase2(self, p, ref_p):
        self._check_inference()
        out, p_c8 = self.phase2_c8(ops.to_nhwc(p), self._ref_c8(ref_p))
        return out, ops.as_nchw(ops.from_c8(p_c8, _lib.NHWC))
