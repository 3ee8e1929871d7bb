"""Weight packer: reference ``state_dict`` tensors -> the layouts the HIP kernels consume.

* conv / linear weights OIHW -> ``[Cout][Kpad]`` with k = (r*S+s)*Cin_pad + ci (arseg_pack_conv_weight_host)
* BatchNorm (eval) folded into a per-channel scale/bias together with the conv bias (arseg_fold_bn_host)
* depthwise 3x3 weights of CReFF ``[C][1][3][3]`` -> ``[9][C]`` (arseg_pack_dw3x3_host)

The arithmetic is done by the host-side entry points of libarseg_hip.so on CPU buffers; this module
only moves the results to the device.  Done once per model (first forward / after load_state_dict).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import check


def _np(t) -> np.ndarray:
    return np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float32)


def _hp(a: Optional[np.ndarray]):
    return ctypes.c_void_p(0 if a is None else a.ctypes.data)


class PackedConv:
    """A conv (or linear) layer with its folded epilogue, resident on ``device``."""

    def __init__(self, weight, conv_bias=None, bn=None, stride=1, pad=0, dil=1, act=_lib.ACT_NONE, slope=0.0, device="cuda",
                 bn_eps=1e-5):
        lib = _lib.load()
        w = _np(weight)
        if w.ndim == 2:                      # nn.Linear == 1x1 conv on a 1x1 image
            w = w[:, :, None, None]
        cout, cin, R, S = w.shape
        self._w_oihw = w                     # host copy (fp32 OIHW) for the lazily built 16-bit packing
        self.cout, self.cin, self.R, self.S = cout, cin, R, S
        self.cin_pad = (cin + 3) // 4 * 4
        self.stride, self.pad, self.dil, self.act, self.slope = int(stride), int(pad), int(dil), int(act), float(slope)
        kpad = lib.arseg_packed_k(self.cin_pad, R, S)
        packed = np.empty((cout, kpad), dtype=np.float32)
        check(lib.arseg_pack_conv_weight_host(_hp(w), cout, cin, R, S, self.cin_pad, _hp(packed)), "pack_conv_weight")
        self.w = torch.from_numpy(packed).to(device)
        # the same weights in the split-fp16 operand format of ARSEG_MATH_F16X3 (+ the per-channel power-of-two factor)
        h3 = np.empty((cout, kpad), dtype=np.float32)
        mul_inv = np.empty(cout, dtype=np.float32)
        check(lib.arseg_split_weight_f16x3_host(_hp(packed), cout, kpad, _hp(h3), _hp(mul_inv)), "split_weight_f16x3")
        self.w_h3 = torch.from_numpy(h3).to(device)
        # Winograd F(4x4,3x3) weights for the convs it applies to (3x3, stride 1, pad == dilation, wide enough channels)
        self.wino_u = None
        if R == 3 and S == 3 and self.stride == 1 and self.pad == self.dil and cin % 32 == 0 and cin >= 64 and cout % 4 == 0:
            u = np.empty((36, cout, cin), dtype=np.float32)
            check(lib.arseg_wino43_pack_weight_host(_hp(w), cout, cin, _hp(u)), "wino43_pack_weight")
            self.wino_u = torch.from_numpy(u).to(device)
            e = 5 - np.frexp(np.maximum(np.abs(u).max(axis=(0, 2)), 1e-30))[1]          # per channel, over the 36 slices
            u_h3 = np.empty_like(u)
            us = np.ascontiguousarray(np.ldexp(u, e[None, :, None].astype(np.int32)), dtype=np.float32)
            check(lib.arseg_split_weight_f16x3_host(_hp(us), 36 * cout, cin, _hp(u_h3), _hp(None)), "split_weight_f16x3")
            self.wino_u_h3 = torch.from_numpy(u_h3).to(device)
            wino_mul_inv = np.ldexp(np.float32(1.0), -e).astype(np.float32)
        cb = None if conv_bias is None else _np(conv_bias)
        if bn is not None:
            gamma, beta, mean, var = (_np(t) for t in bn)
            scale = np.empty(cout, dtype=np.float32)
            bias = np.empty(cout, dtype=np.float32)
            check(lib.arseg_fold_bn_host(_hp(gamma), _hp(beta), _hp(mean), _hp(var), ctypes.c_float(bn_eps), _hp(cb), cout,
                                         _hp(scale), _hp(bias)), "fold_bn")
            self.scale = torch.from_numpy(scale).to(device)
            self.bias = torch.from_numpy(bias).to(device)
        else:
            scale = np.ones(cout, dtype=np.float32)
            self.scale = None
            self.bias = None if cb is None else torch.from_numpy(cb).to(device)
        self.scale_h3 = torch.from_numpy(scale * mul_inv).to(device)
        self.wino_scale_h3 = torch.from_numpy(scale * wino_mul_inv).to(device) if self.wino_u is not None else None

    def weights16(self, dtype):
        """(weights packed for the 16-bit storage path [Cout][Kpad16] in ``dtype``, padded input channel count); built on first use.
        Rounded once from the fp32 parameters (round to nearest even); BN stays in the fp32 epilogue (self.scale / self.bias)."""
        cache = self.__dict__.setdefault("_w16", {})
        if dtype not in cache:
            lib = _lib.load()
            cin_pad = (self.cin + 7) // 8 * 8
            kpad = lib.arseg_packed_k16(cin_pad, self.R, self.S)
            packed = np.empty((self.cout, kpad), dtype=np.uint16)
            code = _lib.DT_BF16 if dtype == torch.bfloat16 else _lib.DT_F16
            check(lib.arseg_pack_conv_weight16_host(_hp(self._w_oihw), self.cout, self.cin, self.R, self.S, cin_pad, code,
                                                    ctypes.c_void_p(packed.ctypes.data)), "pack_conv_weight16")
            cache[dtype] = (torch.from_numpy(packed.view(np.int16)).to(self.w.device).view(dtype), cin_pad)
        return cache[dtype]

    def taps(self):
        """The nine taps of a 3x3 conv stacked along the output channels of ONE 1x1 conv (row t*Cout + co = W[co][.][t//3][t%3], no
        epilogue): the low-resolution GEMM of the tap-decomposed conv-after-upsample route (arseg_upconv3x3_tap_gather_fwd).  Built on
        first use."""
        if "_taps" not in self.__dict__:
            if self.R != 3 or self.S != 3:
                raise _lib.ArsegError("taps(): 3x3 convs only")
            w9 = np.ascontiguousarray(self._w_oihw.transpose(2, 3, 0, 1).reshape(9 * self.cout, self.cin, 1, 1))
            self._taps = PackedConv(w9, None, None, 1, 0, 1, _lib.ACT_NONE, 0.0, self.w.device)
        return self._taps

    @staticmethod
    def from_modules(conv, bn=None, act=_lib.ACT_NONE, slope=0.0, device="cuda"):
        """conv: nn.Conv2d or nn.Linear; bn: nn.BatchNorm2d or None."""
        is_conv = conv.weight.dim() == 4
        return PackedConv(conv.weight, conv.bias,
                          None if bn is None else (bn.weight, bn.bias, bn.running_mean, bn.running_var),
                          stride=conv.stride[0] if is_conv else 1, pad=conv.padding[0] if is_conv else 0,
                          dil=conv.dilation[0] if is_conv else 1, act=act, slope=slope, device=device,
                          bn_eps=1e-5 if bn is None else bn.eps)


class PackedAttention:
    """The three depthwise 3x3 convs of MyAttention (model/attention.py:161-164), tap-major."""

    def __init__(self, attn_module, device="cuda"):
        lib = _lib.load()

        def dw(conv):
            w = _np(conv.weight)
            C = w.shape[0]
            out = np.empty((9, C), dtype=np.float32)
            check(lib.arseg_pack_dw3x3_host(_hp(w), C, _hp(out)), "pack_dw3x3")
            return torch.from_numpy(out).to(device), torch.from_numpy(_np(conv.bias)).to(device)

        self.wq, self.bq = dw(attn_module.lr_query_conv)
        self.wk, self.bk = dw(attn_module.hr_key_conv)
        self.wv, self.bv = dw(attn_module.hr_value_conv)


class PackedHead:
    """final 1x1 classifier (model/pspnet.py:66, model/bisenet.py:211): wf [n_cls][C], bf [n_cls]."""

    def __init__(self, conv, device="cuda"):
        w = _np(conv.weight)
        self.wf = torch.from_numpy(np.ascontiguousarray(w.reshape(w.shape[0], w.shape[1]))).to(device)
        self.bf = torch.from_numpy(_np(conv.bias)).to(device)
