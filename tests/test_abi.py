"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/arseg_hip.h
declares, the Python binding covers exactly that set, the host-side weight packer matches a numpy restatement, the
module mirrors expose the reference's state_dict keys, and nothing silently falls back to the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "arseg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arseg_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from arseg_amd import _lib

    lib = _lib.load()                       # raises if ar-seg_amd/lib/libarseg_hip.so was not built
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in arseg_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == declared, "ctypes binding and header drifted apart"
    assert lib.arseg_version() == _lib.ABI_VERSION == 5
    assert b"ok" in lib.arseg_status_string(0) and b"invalid" in lib.arseg_status_string(-1)


def test_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation happens before any launch, so it is checkable on a CPU-only machine."""
    from arseg_amd import _lib

    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.arseg_local_similar_fwd(null, null, null, 1, 8, 8, 8, 7, 7, null) == _lib.ARSEG_EINVAL
    assert lib.arseg_warp_fwd(null, null, 0, null, 1, 8, 8, 8, 1, 1, null) == _lib.ARSEG_EINVAL
    assert lib.arseg_creff_fwd(*([null] * 8), null, null, null, 0, null, 0, 1, 64, 8, 8, 4, 4, 7, 7, null) == _lib.ARSEG_EINVAL
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.in_ld, d.Cout, d.out_ld, d.res_ld = 1, 8, 8, 24, 24, 8, 8, 8
    d.R, d.S, d.stride, d.pad, d.dil = 3, 3, 1, 1, 1
    ho, wo = ctypes.c_int(), ctypes.c_int()
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == _lib.ARSEG_EUNSUPPORTED   # Cin=24 with 3x3
    d.Cin = d.in_ld = 16
    d.stride, d.pad, d.dil = 2, 4, 4
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (ho.value, wo.value) == ((8 + 8 - 8 - 1) // 2 + 1, (8 + 8 - 8 - 1) // 2 + 1)
    d.split_k = 4
    # K = 9*16 = 144 -> 5 K-steps of 32; 4 requested slices -> 2 steps each -> 3 non-empty slices of fp32 partials
    assert lib.arseg_conv2d_workspace_bytes(ctypes.byref(d)) == 3 * ho.value * wo.value * 8 * 4
    # tile_cfg / math selection: the patch-resident kernel (13..16) covers 3x3 stride-1 pad==dil f16x3 convs with Cin % 32 == 0
    d.split_k, d.stride, d.pad, d.dil, d.tile_cfg, d.math = 0, 1, 1, 1, 15, _lib.MATH_F16X3
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == _lib.ARSEG_EUNSUPPORTED       # Cin = 16
    d.Cin = d.in_ld = 64
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == 0
    d.math = _lib.MATH_F32
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == _lib.ARSEG_EUNSUPPORTED       # f16x3 only
    d.tile_cfg, d.math = 17, _lib.MATH_F32
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == _lib.ARSEG_EUNSUPPORTED       # 8/16-wave tiles: f16x3 only
    d.tile_cfg, d.math = 20, _lib.MATH_F16X3          # (r6) 20..22: the squarer patch tiles -- refused on this 8-pixel-wide map, whose default tile is already narrow
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == _lib.ARSEG_EUNSUPPORTED
    d.tile_cfg = 23
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == _lib.ARSEG_EINVAL
    d.tile_cfg, d.math = 0, 7
    assert lib.arseg_conv_out_hw(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)) == _lib.ARSEG_EINVAL
    # ingest / mergeMotion
    f3 = (ctypes.c_float * 3)(0.5, 0.5, 0.5)
    z3 = (ctypes.c_float * 3)(0.5, 0.0, 0.5)
    assert lib.arseg_frame_u8_to_nhwc4_fwd(null, null, 1, 8, 8, 4, 4, f3, f3, null) == _lib.ARSEG_EINVAL
    one = ctypes.c_void_p(16)                       # non-null, never dereferenced: validation fails first
    assert lib.arseg_frame_u8_to_nhwc4_fwd(one, one, 1, 8, 8, 4, 4, f3, z3, null) == _lib.ARSEG_EINVAL                 # zero std
    assert lib.arseg_merge_motion_workspace_bytes(4, 10, 12) == 5 * 10 * 12 * 16
    assert lib.arseg_merge_motion_fwd(one, one, one, 0, 4, 0, 10, 12, null) == _lib.ARSEG_EWORKSPACE
    assert lib.arseg_merge_motion_fwd(one, one, one, 1 << 20, 4, 4, 10, 12, null) == _lib.ARSEG_EINVAL                 # frame_start >= n_frames
    assert lib.arseg_merge_motion_fwd(null, one, one, 1 << 20, 4, 0, 10, 12, null) == _lib.ARSEG_EINVAL


def test_weight_packer_host_functions():
    from arseg_amd import _lib
    from arseg_amd.packing import _hp

    lib = _lib.load()
    g = np.random.default_rng(0)
    for cout, cin, R, S in ((5, 3, 7, 7), (8, 64, 3, 3), (12, 2560, 1, 1)):
        w = g.standard_normal((cout, cin, R, S)).astype(np.float32)
        cin_pad = (cin + 3) // 4 * 4
        kpad = lib.arseg_packed_k(cin_pad, R, S)
        assert kpad % 32 == 0 and kpad >= R * S * cin_pad
        out = np.full((cout, kpad), 7.0, np.float32)
        assert lib.arseg_pack_conv_weight_host(_hp(w), cout, cin, R, S, cin_pad, _hp(out)) == 0
        want = np.zeros((cout, R * S, cin_pad), np.float32)
        want[:, :, :cin] = w.transpose(0, 2, 3, 1).reshape(cout, R * S, cin)
        assert np.array_equal(out[:, :R * S * cin_pad], want.reshape(cout, -1)) and not out[:, R * S * cin_pad:].any()
    C = 16
    gamma, beta, mean = (g.standard_normal(C).astype(np.float32) for _ in range(3))
    var = g.uniform(0.5, 1.5, C).astype(np.float32)
    cb = g.standard_normal(C).astype(np.float32)
    sc, bi = np.empty(C, np.float32), np.empty(C, np.float32)
    assert lib.arseg_fold_bn_host(_hp(gamma), _hp(beta), _hp(mean), _hp(var), ctypes.c_float(1e-5), _hp(cb), C, _hp(sc), _hp(bi)) == 0
    x = g.standard_normal(C).astype(np.float32)                      # a conv output (without bias)
    want = torch.nn.functional.batch_norm(torch.from_numpy(x + cb)[None, :, None, None], torch.from_numpy(mean), torch.from_numpy(var),
                                          torch.from_numpy(gamma), torch.from_numpy(beta), False, 0.0, 1e-5).flatten().numpy()
    assert np.abs(x * sc + bi - want).max() <= 1e-5
    dw = g.standard_normal((C, 1, 3, 3)).astype(np.float32)
    o = np.empty((9, C), np.float32)
    assert lib.arseg_pack_dw3x3_host(_hp(dw), C, _hp(o)) == 0
    assert np.array_equal(o, dw.reshape(C, 9).T)


def test_packed_conv_taps_layout_and_oracle():
    """PackedConv.taps(): row t*Cout + co of the stacked 1x1 weights = W[co, :, t//3, t%3] (the layout include/arseg_hip.h documents for
    arseg_upconv3x3_tap_gather_fwd), and the identity the route rests on, checked on the CPU with torch ops:
    conv3x3(Up(x)) == sum_t shift_t(Up(W_t x)) with zero padding of the UPSAMPLED image."""
    from arseg_amd import _lib
    from arseg_amd.packing import PackedConv

    g = np.random.default_rng(5)
    cout, cin = 8, 12
    w = g.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    pc = PackedConv(torch.from_numpy(w), None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, "cpu")
    pt = pc.taps()
    assert (pt.cout, pt.cin, pt.R, pt.S, pt.stride, pt.pad) == (9 * cout, cin, 1, 1, 1, 0)
    rows = pt.w.numpy()[:, :cin]
    for t in range(9):
        assert np.array_equal(rows[t * cout:(t + 1) * cout], w[:, :, t // 3, t % 3])
    x = torch.from_numpy(g.standard_normal((2, cin, 5, 7)).astype(np.float32)).double()
    up = lambda a: torch.nn.functional.interpolate(a, scale_factor=2.0, mode="bilinear", align_corners=False)
    want = torch.nn.functional.conv2d(up(x), torch.from_numpy(w).double(), padding=1)
    z = torch.nn.functional.conv2d(x, torch.from_numpy(rows.reshape(9 * cout, cin, 1, 1)).double())          # the low-resolution GEMM
    zu = torch.nn.functional.pad(up(z), (1, 1, 1, 1))                                                        # zero outside the upsampled image
    H, W = want.shape[-2:]
    got = sum(zu[:, t * cout:(t + 1) * cout, t // 3:t // 3 + H, t % 3:t % 3 + W] for t in range(9))
    assert float((got - want).abs().max()) <= 1e-12


def test_split_weight_f16x3_host():
    """hi + lo (two fp16 per weight, row pre-scaled by a power of two) reproduces the fp32 weight to ~2^-21 of the row maximum."""
    from arseg_amd import _lib

    def _hp(a):
        return ctypes.c_void_p(a.ctypes.data)

    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(17))
    co, k = 6, 96
    w = (g.standard_normal((co, k)) * np.array([1e-4, 1e-2, 1.0, 40.0, 3e-7, 0.0])[:, None]).astype(np.float32)
    out, inv = np.empty((co, k), np.float32), np.empty(co, np.float32)
    assert lib.arseg_split_weight_f16x3_host(_hp(w), co, k, _hp(out), _hp(inv)) == 0
    h = out.view(np.float16).reshape(co, k // 32, 2, 32).astype(np.float64)
    rec = (h[:, :, 0, :] + h[:, :, 1, :]).reshape(co, k) * inv[:, None].astype(np.float64)
    scale = np.maximum(np.abs(w).max(axis=1, keepdims=True), 1e-30)
    assert (np.abs(rec - w) / scale).max() <= 2.0 ** -20
    assert np.all(np.log2(inv) == np.round(np.log2(inv)))                  # exact powers of two (undone in the epilogue scale)
    assert lib.arseg_split_weight_f16x3_host(_hp(w), co, 40, _hp(out), _hp(inv)) == _lib.ARSEG_EINVAL    # K not a multiple of 32
    out2 = np.empty((co, k), np.float32)
    assert lib.arseg_split_weight_f16x3_host(_hp(w), co, k, _hp(out2), ctypes.c_void_p(0)) == 0           # unscaled variant
    h2 = out2.view(np.float16).reshape(co, k // 32, 2, 32).astype(np.float64)
    assert np.abs((h2[:, :, 0, :] + h2[:, :, 1, :]).reshape(co, k)[2] - w[2]).max() <= 2.0 ** -20 * np.abs(w[2]).max()


def test_module_mirrors_have_the_reference_state_dict(manifest):
    from arseg_amd.model import BiSeNetV1, BiSeNetV1WithFuse, MyAttention, PSPNet, PSPNetWithFuse, pspnet_semseg

    ctors = {
        "PSPNet": lambda: PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18"),
        "PSPNetWithFuse": lambda: PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256,
                                                 backend="resnet18", atten_k=7),
        "BiSeNetV1": lambda: BiSeNetV1(n_classes=12, backend="resnet18"),
        "BiSeNetV1WithFuse": lambda: BiSeNetV1WithFuse(n_classes=12, backend="resnet18"),
        "MyAttention64": lambda: MyAttention(64, kW=7, kH=7),
        "SemsegPSPNet": lambda: pspnet_semseg.PSPNet(bins=(1, 2, 3, 6), classes=19, feat_dim=512, layers=18),
        "SemsegPSPNetWithFuse": lambda: pspnet_semseg.PSPNetWithFuse(bins=(1, 2, 3, 6), classes=19, feat_dim=512, layers=18),
    }
    for name, ctor in ctors.items():
        m = ctor()
        got = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        assert got == manifest[name]["keys"], name                   # same keys, shapes and order as the reference module
        wrapped = torch.nn.DataParallel(m)                           # checkpoints carry the 'module.' prefix (evaluation.py:41-46)
        wrapped.load_state_dict({"module." + k: v for k, v in m.state_dict().items()})
    b = ctors["BiSeNetV1WithFuse"]()
    assert b.feat_conv_out is b.conv_out.conv and b.final_conv is b.conv_out.conv_out and b.out_upsample is b.conv_out.up
    s = ctors["SemsegPSPNetWithFuse"]()
    assert s.final_conv is s.cls[4]


def test_no_cpu_fallback():
    from arseg_amd import _lib, ops
    from arseg_amd.model import MyAttention, PSPNet
    from arseg_amd.evaluation import warpFeature

    with pytest.raises(_lib.ArsegError):
        ops.local_similar(torch.zeros(1, 4, 8, 8), torch.zeros(1, 4, 8, 8), 7, 7)
    with pytest.raises(_lib.ArsegError):
        warpFeature(torch.zeros(1, 4, 8, 8), torch.zeros(1, 8, 8, 2))
    with pytest.raises(_lib.ArsegError):
        MyAttention(8, kW=7, kH=7).eval()(torch.zeros(1, 8, 8, 8), torch.zeros(1, 8, 4, 4))
    net = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
    with pytest.raises(_lib.ArsegError):                              # training mode: inference-only
        net(torch.zeros(1, 3, 32, 32))
    with pytest.raises(_lib.ArsegError):                              # eval mode, CPU tensors
        net.eval()(torch.zeros(1, 3, 32, 32))


def test_synthetic_clip_matches_the_reference_mv_format():
    from arseg_amd import synth

    c = synth.make_clip(0, 64, 96, gop=4)
    assert c["frames"].shape == (4, 3, 64, 96) and c["frames"].dtype == np.float32
    mv = c["mv"]
    assert mv.shape == (4, 64, 96, 2) and mv.dtype == np.int16
    assert not mv[0].any() and (mv % 4 == 0).all() and np.abs(mv).max() <= 150 * 4      # integer-pel, clamped, keyframe has no motion
    c2 = synth.make_clip(0, 64, 96, gop=4)
    assert np.array_equal(c["frames"], c2["frames"]) and np.array_equal(mv, c2["mv"])     # seeded


def test_reference_attention_imports_with_shim():
    """INTEGRATION.md level 1: with shims/ in front of sys.path the UNMODIFIED reference model/attention.py imports, its
    ``from localAttention import ...`` resolving to the HIP-backed shim and its own ``model`` package staying the reference's."""
    import subprocess
    import sys

    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "model")):
        pytest.skip("reference tree not present (GPU box)")
    code = (
        "import sys; sys.dont_write_bytecode = True\n"
        f"sys.path.insert(0, {ref!r}); sys.path.insert(0, {os.path.join(ROOT, 'shims')!r})\n"
        "import localAttention, model.attention as A\n"
        f"assert localAttention.__file__.startswith({os.path.join(ROOT, 'shims')!r}), localAttention.__file__\n"
        f"assert A.__file__.startswith({ref!r}), A.__file__\n"
        "assert A.similar_forward is localAttention.similar_forward and hasattr(A, 'MyAttention')\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_header_is_plain_c_and_links(tmp_path):
    """include/arseg_hip.h must be consumable from C (the boundary is a C ABI: cgo / JNI / ctypes style bindings): a C99 translation
    unit including it compiles with -pedantic, links against the library and calls host-only entry points."""
    import shutil
    import subprocess

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "use.c"
    src.write_text('#include "arseg_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { arseg_conv_desc d; (void)d;\n'
                   '  if (arseg_conv2d_find(0,0,0,0,0,0,0,0,0,0,0,0,0,0) != ARSEG_EINVAL) return 2;\n'
                   '  if (arseg_packed_k(64, 3, 3) != 576) return 3;\n'
                   '  printf("%d %s\\n", arseg_version(), arseg_status_string(ARSEG_EUNSUPPORTED)); return 0; }\n')
    libdir = os.path.join(ROOT, "ar-seg_amd", "lib")
    exe = tmp_path / "use"
    subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-larseg_hip", f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) >= 1
