"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (``ar-seg_amd/``).

CPU restatement (plain PyTorch fp32 functional ops + explicit loops over the attention
window) of the AR-Seg LR-branch inference hot path, SURVEY.md section 8 rows a1-a22.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this file, and only as the checker / the timed CPU baseline.

Parity status
-------------
* Everything except the ``localAttention`` pair is pinned against the reference itself:
  ``tests/golden/make_golden.py`` imports ``/root/reference/model/*.py`` and
  ``evaluation.py`` (CPU, with import shims for absent third-party modules), runs them on
  seeded inputs and commits the input/output vectors under ``tests/golden/``;
  ``tests/test_oracle_golden.py`` checks every function here against those vectors.
* ``localAttention`` (github.com/zzd1992/Image-Local-Attention, pinned only as ``@master`` in
  the reference's requirements.txt:7, NOT vendored) is absent from ``/root/reference`` and
  cannot be built here (CUDA).  ``local_weighting`` is pinned against the reference's own
  in-tree CPU restatement ``f_weighting_cpu`` (model/attention.py:75-85).  For
  ``local_similar`` the reference holds no runnable implementation, test or golden vector
  (``f_similar_cpu`` model/attention.py:55-73 is broken debug code): its contract is taken
  from the layout comments there (zero-padded ``nn.Unfold``, tap index ``i = dy*kW + dx``,
  output ``[N,H,W,kH*kW]``) and from the published algorithm of the third-party op --
  **parity unpinned** for that one function with respect to the third-party CUDA code.

All tensors are NCHW float32 unless stated.  ``sd`` is a reference ``state_dict`` (keys
without the ``module.`` prefix) mapping to torch tensors.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-5  # nn.BatchNorm2d default, used everywhere in the reference


def strip_module_prefix(sd: SD) -> SD:
    """Checkpoints are saved from nn.DataParallel (evaluation.py:41-46): keys start with 'module.'."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------
# a1  warpFeature                                                         evaluation.py:61-87
# ----------------------------------------------------------------------------------------------
def warp_feature(feature: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """feature [B,C,H,W] f32, flow [B,H,W,2] (dx,dy in feature pixels; f32 or f64).

    Base grid (x,y) + flow (evaluation.py:67-77; float32 + float64 promotes to float64),
    normalised with the align_corners=True formula 2*g/(W-1)-1 (80-81), cast to f32 (83) and
    fed to grid_sample with its defaults bilinear / zeros / align_corners=False (85).
    """
    B, C, H, W = feature.shape
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(B, H, W)
    ys = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(B, H, W)
    gx = xs + flow[..., 0]
    gy = ys + flow[..., 1]
    gx = 2.0 * gx / max(W - 1, 1) - 1.0
    gy = 2.0 * gy / max(H - 1, 1) - 1.0
    grid = torch.stack((gx, gy), dim=-1).float()
    return F.grid_sample(feature, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


# ----------------------------------------------------------------------------------------------
# a2  motion-vector resize                                               evaluation.py:176-180
# ----------------------------------------------------------------------------------------------
def mv_resize(flow: torch.Tensor, Hp: int, Wp: int) -> torch.Tensor:
    """flow [B,H,W,2] (f64 from int16/4, dataset/camvid.py:625) -> [B,Hp,Wp,2], same dtype.

    Both components are scaled by Hp/H (evaluation.py:178 -- also the x component), then
    bilinear align_corners=True resampling (179).
    """
    f = flow.permute(0, 3, 1, 2)
    f = f * Hp / f.shape[-2]
    f = F.interpolate(f, [Hp, Wp], mode="bilinear", align_corners=True)
    return f.permute(0, 2, 3, 1)


def mv_from_int16(mv_q: torch.Tensor) -> torch.Tensor:
    """int16 quarter-pel [..,H,W,2] -> float64 pixels (dataset/camvid.py:625, cityscapes.py:283)."""
    return mv_q.to(torch.float64) / 4


# ----------------------------------------------------------------------------------------------
# a3  decoded-frame downscale                                   evaluation.py:115-117,186-188
# ----------------------------------------------------------------------------------------------
def to_tensor_normalize(img_u8_hwc, mean, std) -> torch.Tensor:
    """transforms.ToTensor() + transforms.Normalize(mean, std) (dataset/camvid.py:503-506) restated with torch ops (torchvision
    is not installed here): uint8 HWC -> float CHW / 255, then (x - mean) / std per channel, fp32.  [N,H,W,3] -> [N,3,H,W]."""
    x = torch.as_tensor(img_u8_hwc).permute(0, 3, 1, 2).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return x.sub(m).div(s)


def downscale(imgs: torch.Tensor, scale: float) -> torch.Tensor:
    H, W = imgs.shape[-2:]
    return F.interpolate(imgs, [int(H * scale), int(W * scale)], mode="bilinear", align_corners=True)


# ----------------------------------------------------------------------------------------------
# a5-a7  localAttention pair + softmax             model/attention.py:13-53 (call sites 199-207)
# ----------------------------------------------------------------------------------------------
_ROWS = 16   # row strip: keeps the working set of one strip in cache (a whole 64x512x1024 plane set is 134 MB)


_C_THREADS = None      # None: the torch strip loops below; an int n >= 0: oracle/local_attn_ref.c's OpenMP variants on n threads (0 = all)


def use_c_local_attention(nthreads=0):
    """Route local_similar / local_weighting through the plain-C restatement's multi-threaded variants (oracle_local_*_mt, bit-equal to its
    scalar functions): the torch strip loops stop scaling near 16 threads, the C row loops use every host core -- what bench.py's
    cpu_baseline leg times at os.cpu_count() threads.  ``None`` switches back.  Returns the previous setting."""
    global _C_THREADS
    prev, _C_THREADS = _C_THREADS, nthreads
    return prev


def local_similar(q: torch.Tensor, k: torch.Tensor, kH: int, kW: int) -> torch.Tensor:
    """S[n,y,x,dy*kW+dx] = sum_c q[n,c,y,x] * k[n,c,y+dy-kH//2,x+dx-kW//2]; taps outside the
    image contribute 0 (zero padding of the unfold, attention.py:56-58).  No 1/sqrt(C) scale.
    Row-tiled: a naive F.unfold materialises C*kH*kW*H*W floats (6.6 GB at 64x512x1024)."""
    N, C, H, W = q.shape
    if _C_THREADS is not None:
        from . import c_ref
        return torch.from_numpy(c_ref.local_similar_mt(q.detach().numpy(), k.detach().numpy(), kH, kW, _C_THREADS))
    kp = F.pad(k, (kW // 2, kW // 2, kH // 2, kH // 2))
    out = q.new_empty(N, H, W, kH * kW)
    for y0 in range(0, H, _ROWS):
        y1 = min(H, y0 + _ROWS)
        qs = q[:, :, y0:y1]
        for dy in range(kH):
            for dx in range(kW):
                out[:, y0:y1, :, dy * kW + dx] = (qs * kp[:, :, y0 + dy:y1 + dy, dx:dx + W]).sum(dim=1)
    return out


def local_weighting(v: torch.Tensor, w: torch.Tensor, kH: int, kW: int) -> torch.Tensor:
    """O[n,c,y,x] = sum_i v[n,c,y+dy_i,x+dx_i] * w[n,y,x,i], zero padded (attention.py:75-85)."""
    N, C, H, W = v.shape
    if _C_THREADS is not None:
        from . import c_ref
        return torch.from_numpy(c_ref.local_weighting_mt(v.detach().numpy(), w.detach().numpy(), kH, kW, _C_THREADS))
    vp = F.pad(v, (kW // 2, kW // 2, kH // 2, kH // 2))
    out = torch.empty_like(v)
    for y0 in range(0, H, _ROWS):
        y1 = min(H, y0 + _ROWS)
        acc = torch.zeros_like(v[:, :, y0:y1])
        for dy in range(kH):
            for dx in range(kW):
                acc += vp[:, :, y0 + dy:y1 + dy, dx:dx + W] * w[:, y0:y1, :, dy * kW + dx].unsqueeze(1)
        out[:, :, y0:y1] = acc
    return out


# ----------------------------------------------------------------------------------------------
# a4  MyAttention (CReFF fusion)                                    model/attention.py:157-213
# ----------------------------------------------------------------------------------------------
def my_attention(sd: SD, prefix: str, hr_feat: torch.Tensor, lr_feat: torch.Tensor, kH: int = 7, kW: int = 7,
                 return_parts: bool = False):
    C = hr_feat.shape[1]
    H, W = hr_feat.shape[-2:]
    lr_up = F.interpolate(lr_feat, (H, W), mode="bilinear", align_corners=True)                   # :191

    def dw(name, x):  # depthwise 3x3 + bias, groups = feat_dim (:161-164)
        return F.conv2d(x, sd[f"{prefix}{name}.weight"], sd[f"{prefix}{name}.bias"], padding=1, groups=C)

    v = dw("hr_value_conv", hr_feat)                                                                # :194
    k = dw("hr_key_conv", hr_feat)                                                                  # :196
    q = dw("lr_query_conv", lr_up)                                                                  # :197
    s = local_similar(q, k, kH, kW)                                                                 # :199
    wgt = torch.softmax(s, dim=3)                                                                   # :203
    a = local_weighting(v, wgt, kH, kW)                                                             # :207
    out = lr_up + a                                                                                 # :210
    if return_parts:
        return out, dict(lr_up=lr_up, q=q, k=k, v=v, s=s, w=wgt, a=a)
    return out


# ----------------------------------------------------------------------------------------------
# shared conv helpers
# ----------------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, BN_EPS)


def _conv(sd: SD, p: str, x: torch.Tensor, stride=1, padding=0, dilation=1) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding, dilation=dilation)


# ----------------------------------------------------------------------------------------------
# a8  dilated ResNet-18 feature extractor                       model/extractors.py:35-66,108-158
# ----------------------------------------------------------------------------------------------
def _basic_block_ext(sd: SD, p: str, x: torch.Tensor, stride: int, dilation: int) -> torch.Tensor:
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, dilation, dilation)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, dilation, dilation))
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride))
    return F.relu(out + res)


def resnet18_dilated(sd: SD, p: str, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (layer4 out 512ch, layer3 out 256ch), both stride 8 (extractors.py:146-158).

    ``_make_layer`` does not forward ``dilation`` to the first block of a layer
    (extractors.py:139 vs 142): layer3 blocks are dilated [1,2], layer4 [1,4].
    """
    x = F.relu(_bn(sd, p + "bn1", _conv(sd, p + "conv1", x, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for name, stride, dil in (("layer1", 1, 1), ("layer2", 2, 1), ("layer3", 1, 2), ("layer4", 1, 4)):
        x = _basic_block_ext(sd, f"{p}{name}.0", x, stride, 1)
        x = _basic_block_ext(sd, f"{p}{name}.1", x, 1, dil)
        if name == "layer3":
            x3 = x
    return x, x3


# ----------------------------------------------------------------------------------------------
# a9-a12  PSPNet-18 (CamVid)                                               model/pspnet.py:14-231
# ----------------------------------------------------------------------------------------------
def psp_module(sd: SD, p: str, feats: torch.Tensor, sizes=(1, 2, 3, 6)) -> torch.Tensor:
    h, w = feats.shape[-2:]
    priors = []
    for i, s in enumerate(sizes):
        t = F.adaptive_avg_pool2d(feats, (s, s))
        t = _conv(sd, f"{p}stages.{i}.1", t)                                   # 1x1, no bias (:24)
        priors.append(F.interpolate(t, (h, w), mode="bilinear", align_corners=False))  # F.upsample default (:29)
    priors.append(feats)
    return F.relu(_conv(sd, p + "bottleneck", torch.cat(priors, 1)))           # :30-31


def psp_upsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    h, w = 2 * x.shape[2], 2 * x.shape[3]
    x = F.interpolate(x, (h, w), mode="bilinear", align_corners=False)         # :45
    x = _bn(sd, p + "conv.1", _conv(sd, p + "conv.0", x, 1, 1))
    return F.prelu(x, sd[p + "conv.2.weight"])


def pspnet_trunk(sd: SD, x: torch.Tensor, sizes=(1, 2, 3, 6)):
    """feats -> psp -> up_1..3 (dropouts are identity in eval); returns (aux logits, p)."""
    f, class_f = resnet18_dilated(sd, "feats.", x)
    p = psp_module(sd, "psp.", f, sizes)
    p = psp_upsample(sd, "up_1.", p)
    p = psp_upsample(sd, "up_2.", p)
    p = psp_upsample(sd, "up_3.", p)
    aux = F.adaptive_max_pool2d(class_f, (1, 1)).view(-1, class_f.size(1))
    aux = F.linear(F.relu(F.linear(aux, sd["classifier.0.weight"], sd["classifier.0.bias"])),
                   sd["classifier.2.weight"], sd["classifier.2.bias"])
    return aux, p


def pspnet_forward(sd: SD, x: torch.Tensor, sizes=(1, 2, 3, 6)):
    """PSPNet.forward / PSPNetWithFuse.forward(mode='normal') (pspnet.py:76-100,166-189)."""
    H, W = x.shape[-2:]
    aux, p = pspnet_trunk(sd, x, sizes)
    out = _conv(sd, "final_conv", p)
    out = F.interpolate(out, (H, W), mode="bilinear", align_corners=True)
    out = F.log_softmax(out, dim=1)                                            # nn.LogSoftmax() implicit dim -> 1 for 4-D
    return out, aux, p


def pspnet_fuse_phase1(sd: SD, x: torch.Tensor, sizes=(1, 2, 3, 6)):
    return pspnet_trunk(sd, x, sizes)                                          # pspnet.py:198-217


def pspnet_fuse_phase2(sd: SD, p: torch.Tensor, ref_p: torch.Tensor, k: int = 7):
    H, W = ref_p.shape[-2:]
    p = my_attention(sd, "fuse_attention.", ref_p, p, k, k)                    # pspnet.py:223
    out = _conv(sd, "final_conv", p)
    out = F.interpolate(out, (H, W), mode="bilinear", align_corners=True)
    return F.log_softmax(out, dim=1), p


# ----------------------------------------------------------------------------------------------
# f1  PSPNet-18 (Cityscapes)                                        model/pspnet_semseg.py:12-250
# ----------------------------------------------------------------------------------------------
def _basic_block_semseg(sd: SD, p: str, x: torch.Tensor, stride: int, dil1: int, dil2: int) -> torch.Tensor:
    """extractors.BasicBlock whose conv2 was re-dilated after construction (pspnet_semseg.py:59-68): conv1 keeps the
    extractor's dilation (1 in the first block of a layer, the layer's in the second), conv2 gets the layer's."""
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, dil1, dil1)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, dil2, dil2))
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride))
    return F.relu(out + res)


def semseg_phase1(sd: SD, x: torch.Tensor, bins=(1, 2, 3, 6)):
    """PSPNetWithFuse.forward_phase1 (pspnet_semseg.py:219-231): -> (x_tmp = layer3 output, p = cls[:-1](ppm(layer4)))."""
    x = F.relu(_bn(sd, "layer0.1", _conv(sd, "layer0.0", x, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for name, stride, dil in (("layer1", 1, 1), ("layer2", 2, 1), ("layer3", 1, 2), ("layer4", 1, 4)):
        x = _basic_block_semseg(sd, f"{name}.0", x, stride, 1, dil if name in ("layer3", "layer4") else 1)
        x = _basic_block_semseg(sd, f"{name}.1", x, 1, dil, dil)
        if name == "layer3":
            x_tmp = x
    out = [x]
    for i, b in enumerate(bins):                                                        # PPM.forward (:24-30)
        t = F.adaptive_avg_pool2d(x, b)
        t = F.relu(_bn(sd, f"ppm.features.{i}.2", _conv(sd, f"ppm.features.{i}.1", t)))
        out.append(F.interpolate(t, x.shape[2:], mode="bilinear", align_corners=True))
    p = F.relu(_bn(sd, "cls.1", _conv(sd, "cls.0", torch.cat(out, 1), 1, 1)))           # cls[:-1]; Dropout2d = identity
    return x_tmp, p


def semseg_forward(sd: SD, x: torch.Tensor, bins=(1, 2, 3, 6), zoom_factor: int = 8):
    """PSPNetWithFuse.forward(x, mode='normal') (pspnet_semseg.py:186-217): (logits, aux logits, p); PSPNet.forward is [0]."""
    h, w = x.shape[2:]
    x_tmp, p = semseg_phase1(sd, x, bins)
    out = _conv(sd, "cls.4", p)
    aux = _conv(sd, "aux.4", F.relu(_bn(sd, "aux.1", _conv(sd, "aux.0", x_tmp, 1, 1))))
    if zoom_factor != 1:
        out = F.interpolate(out, size=(h, w), mode="bilinear", align_corners=True)
        aux = F.interpolate(aux, size=(h, w), mode="bilinear", align_corners=True)
    return out, aux, p


def semseg_phase2(sd: SD, p: torch.Tensor, ref_p: torch.Tensor, k: int = 7):
    """PSPNetWithFuse.forward_phase2 (pspnet_semseg.py:233-247): raw logits at feature resolution, fused feature."""
    p = my_attention(sd, "fuse_attention.", ref_p, p, k, k)
    return _conv(sd, "final_conv", p), p


# ----------------------------------------------------------------------------------------------
# a13-a21  BiSeNetV1-18                                                    model/bisenet.py
# ----------------------------------------------------------------------------------------------
def _cbr(sd: SD, p: str, x: torch.Tensor, stride=1, padding=1) -> torch.Tensor:
    """ConvBNReLU (bisenet.py:162-180)."""
    return F.relu(_bn(sd, p + ".bn", _conv(sd, p + ".conv", x, stride, padding)))


def _basic_block_bise(sd: SD, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    r = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1)))
    r = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", r, 1, 1))
    sc = x
    if (p + ".downsample.0.weight") in sd:
        sc = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride))
    return F.relu(sc + r)


def resnet18_s32(sd: SD, p: str, x: torch.Tensor):
    """bisenet.Resnet18.forward (bisenet.py:84-94): feat8, feat16, feat32."""
    x = F.relu(_bn(sd, p + "bn1", _conv(sd, p + "conv1", x, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2), ("layer4", 2)):
        x = _basic_block_bise(sd, f"{p}{name}.0", x, stride)
        x = _basic_block_bise(sd, f"{p}{name}.1", x, 1)
        feats.append(x)
    return feats[1], feats[2], feats[3]


def arm(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """AttentionRefinementModule (bisenet.py:243-260)."""
    feat = _cbr(sd, p + "conv", x)
    att = feat.mean(dim=(2, 3), keepdim=True)
    att = _bn(sd, p + "bn_atten", _conv(sd, p + "conv_atten", att)).sigmoid()
    return feat * att


def context_path(sd: SD, p: str, x: torch.Tensor):
    """ContextPath.forward (bisenet.py:289-306)."""
    feat8, feat16, feat32 = resnet18_s32(sd, p + "resnet.", x)
    avg = _cbr(sd, p + "conv_avg", feat32.mean(dim=(2, 3), keepdim=True), 1, 0)
    f32 = arm(sd, p + "arm32.", feat32) + avg
    f32 = F.interpolate(f32, scale_factor=2.0, mode="nearest")                          # nn.Upsample(scale_factor=2.) (:284)
    f32 = F.interpolate(f32, list(feat16.shape[-2:]), mode="bilinear", align_corners=True)  # :298
    f32 = _cbr(sd, p + "conv_head32", f32)
    f16 = arm(sd, p + "arm16.", feat16) + f32
    f16 = F.interpolate(f16, scale_factor=2.0, mode="nearest")
    f16 = _cbr(sd, p + "conv_head16", f16)
    return f16, f32


def spatial_path(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    x = _cbr(sd, p + "conv1", x, 2, 3)
    x = _cbr(sd, p + "conv2", x, 2, 1)
    x = _cbr(sd, p + "conv3", x, 2, 1)
    return _cbr(sd, p + "conv_out", x, 1, 0)


def ffm(sd: SD, p: str, fsp: torch.Tensor, fcp: torch.Tensor) -> torch.Tensor:
    """FeatureFusionModule.forward (bisenet.py:387-399)."""
    feat = _cbr(sd, p + "convblk", torch.cat([fsp, fcp], dim=1), 1, 0)
    att = feat.mean(dim=(2, 3), keepdim=True)
    att = _bn(sd, p + "bn", _conv(sd, p + "conv", att)).sigmoid()
    return feat * att + feat


def bisenet_output(sd: SD, p: str, x: torch.Tensor, up_factor: int) -> torch.Tensor:
    """BiSeNetOutput.forward (bisenet.py:218-222)."""
    x = _conv(sd, p + "conv_out", _cbr(sd, p + "conv", x))
    return F.interpolate(x, scale_factor=float(up_factor), mode="bilinear", align_corners=False)


def _bisenet_trunk(sd: SD, x: torch.Tensor):
    cp8, cp16 = context_path(sd, "cp.", x)
    sp = spatial_path(sd, "sp.", x)
    sp = F.interpolate(sp, list(cp8.shape[-2:]), mode="bilinear", align_corners=True)      # :442
    fuse = ffm(sd, "ffm.", sp, cp8)
    mid = _cbr(sd, "conv_out.conv", fuse)                                                   # feat_conv_out alias (:428)
    return cp8, cp16, mid


def bisenet_forward(sd: SD, x: torch.Tensor, aux_mode: str = "train"):
    """BiSeNetV1.forward / BiSeNetV1WithFuse.forward(mode='normal') (bisenet.py:438-461)."""
    cp8, cp16, mid = _bisenet_trunk(sd, x)
    out = _conv(sd, "conv_out.conv_out", mid)                                               # final_conv alias
    out = F.interpolate(out, scale_factor=8.0, mode="bilinear", align_corners=False)        # out_upsample alias
    if aux_mode == "train":
        return out, bisenet_output(sd, "conv_out16.", cp8, 8), bisenet_output(sd, "conv_out32.", cp16, 16), mid
    if aux_mode == "eval":
        return (out,)
    if aux_mode == "pred":
        return out.argmax(dim=1)
    raise NotImplementedError


def bisenet_fuse_phase1(sd: SD, x: torch.Tensor, aux_mode: str = "train"):
    cp8, cp16, mid = _bisenet_trunk(sd, x)                                                  # bisenet.py:546-561
    if aux_mode == "train":
        return bisenet_output(sd, "conv_out16.", cp8, 8), bisenet_output(sd, "conv_out32.", cp16, 16), mid
    if aux_mode == "eval":
        return mid
    raise NotImplementedError


def bisenet_fuse_phase2(sd: SD, mid: torch.Tensor, ref_p: torch.Tensor, k: int = 7):
    p = my_attention(sd, "fuse_attention.", ref_p, mid, k, k)                               # bisenet.py:569
    out = _conv(sd, "conv_out.conv_out", p)
    return F.interpolate(out, scale_factor=8.0, mode="bilinear", align_corners=False), p


# ----------------------------------------------------------------------------------------------
# a22  logits -> prediction -> confusion matrix -> mIoU                  evaluation.py:201-213
# ----------------------------------------------------------------------------------------------
def eval_tail(logits: torch.Tensor, label: torch.Tensor, n_classes: int, ignore_label: int = 255):
    size = label.shape[-2:]
    logits = F.interpolate(logits, size=size, mode="bilinear", align_corners=True)
    preds = torch.argmax(torch.softmax(logits, dim=1), dim=1)
    keep = label != ignore_label
    hist = torch.bincount(label[keep] * n_classes + preds[keep], minlength=n_classes ** 2)
    return preds, hist.view(n_classes, n_classes).float()


def miou(hist: torch.Tensor) -> torch.Tensor:
    return (hist.diag() / (hist.sum(dim=0) + hist.sum(dim=1) - hist.diag())).mean()


# ----------------------------------------------------------------------------------------------
# one non-keyframe of EvalAlterRes.__call__                              evaluation.py:161-209
# ----------------------------------------------------------------------------------------------
def alter_res_step(kind: str, sd_hr: SD, sd_lr: SD, img: torch.Tensor, ref_img: torch.Tensor, flow: torch.Tensor,
                   scale: float = 0.5, ref_p: torch.Tensor | None = None):
    """kind: 'psp' | 'bise' | 'semseg'.  flow: [1,H,W,2] float64 pixels.  Returns (logits, p, ref_p_warped, ref_p)."""
    if ref_p is None:
        fwd = {"psp": pspnet_forward, "bise": bisenet_forward, "semseg": semseg_forward}[kind]
        ref_p = fwd(sd_hr, ref_img)[-1]                                                                        # :173-174
    Hp, Wp = ref_p.shape[-2:]
    f = mv_resize(flow, Hp, Wp)                                                                               # :177-180
    warped = warp_feature(ref_p, f)                                                                           # :183
    lr = downscale(img, scale)                                                                                # :186-188
    if kind == "psp":
        out_p = pspnet_fuse_phase1(sd_lr, lr)[-1]
        out, p = pspnet_fuse_phase2(sd_lr, out_p, warped)
    elif kind == "semseg":
        out_p = semseg_phase1(sd_lr, lr)[-1]
        out, p = semseg_phase2(sd_lr, out_p, warped)
    else:
        out_p = bisenet_fuse_phase1(sd_lr, lr)[-1]
        out, p = bisenet_fuse_phase2(sd_lr, out_p, warped)
    return out, p, warped, ref_p


# ----------------------------------------------------------------------------------------------
# f4  mergeMotion: chain per-frame codec MVs back to the keyframe
#                                        pre-process/generate_compressed_dataset_camvid.py:6-56
# ----------------------------------------------------------------------------------------------
def merge_motion(flows, frame_start: int = 0):
    """flows: int16 [F+1,H,W,3] (mv_x, mv_y quarter-pel, reference index; entry <= frame_start unused).
    Returns int32 [H,W,F+1,2]: for frames > frame_start the accumulated quarter-pel motion to the keyframe, -1 for frame
    `frame_start` and earlier (the reference never overwrites them).  Restated step by step: intra blocks (ref < 0 or >= 3)
    get zero motion and reference 0 (:20-22); target = pixel + np.round(mv/4) (half to even, :26-27), clipped into the image
    (:33-34); target frame f2 = max(0, f1 - ref - 1) (:28); if the target already has a parent the pixel links to the
    target's parent, otherwise to the target itself (:37-49; the `== 90` branch is dead after the intra masking); finally
    (x, y) positions become quarter-pel displacements (:53-54)."""
    flows = np.array(flows, dtype=np.int16, copy=True)
    F1, H, W, _ = flows.shape
    dp = np.full((H, W, F1, 3), -1, dtype=np.int32)
    k1, j1 = np.meshgrid(np.arange(W), np.arange(H))
    for f1 in range(frame_start + 1, F1):
        flow = flows[f1]
        intra = np.logical_or(flow[..., 2] < 0, flow[..., 2] >= 3)
        flow[intra] = 0
        j2 = np.clip(j1 + np.round(flow[..., 1] / 4).astype(int), 0, H - 1)
        k2 = np.clip(k1 + np.round(flow[..., 0] / 4).astype(int), 0, W - 1)
        f2 = np.maximum(0, f1 - flow[..., 2].astype(int) - 1)
        parent = dp[j2, k2, f2]                                                 # [H,W,3]
        has_parent = parent[..., 2] != -1
        dp[:, :, f1] = np.where(has_parent[..., None], parent, np.stack([k2, j2, f2], axis=-1))
    dp[:, :, 1:, 0] = (dp[:, :, 1:, 0] - k1[..., None]) * 4
    dp[:, :, 1:, 1] = (dp[:, :, 1:, 1] - j1[..., None]) * 4
    return dp[:, :, :, :2]
