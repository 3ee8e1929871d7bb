"""Pins the oracle (oracle/cpu_ref.py, oracle/local_attn_ref.c) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest
import torch

from helpers import maxdiff, sd_from_manifest, t
from oracle import c_ref, cpu_ref

TOL = 1e-5  # oracle vs reference-generated golden


def test_synth_weights_match_manifest_hash(manifest):
    for name, seed in (("PSPNet", 0), ("PSPNetWithFuse", 1), ("BiSeNetV1", 2), ("BiSeNetV1WithFuse", 3), ("MyAttention64", 0)):
        sd = sd_from_manifest(manifest, name, seed)
        h = hashlib.sha256()
        for k, v in sd.items():
            h.update(k.encode())
            h.update(np.ascontiguousarray(v.numpy()).tobytes())
        assert h.hexdigest() == manifest[name]["sha256"], name


@pytest.mark.parametrize("seed", [0, 1])
def test_g1_warp(golden, seed):
    g = golden(f"g1_warp_s{seed}")
    feat = t(g["feat"])
    for k in ("int", "frac", "f32"):
        assert maxdiff(cpu_ref.warp_feature(feat, t(g[f"flow_{k}"])), g[f"out_{k}"]) <= TOL
    zero = torch.zeros(1, 12, 16, 2, dtype=torch.float64)
    assert maxdiff(cpu_ref.warp_feature(feat, zero), g["out_zero"]) <= TOL
    # zero motion is NOT an identity warp (align_corners mismatch)
    assert maxdiff(g["out_zero"], g["feat"]) > 1e-2


@pytest.mark.parametrize("seed", [0, 1])
def test_g2_mv_resize(golden, seed):
    g = golden(f"g2_mvresize_s{seed}")
    flow = cpu_ref.mv_from_int16(t(g["mvq"]))
    for hp, wp in ((4, 6), (32, 48), (5, 7)):
        out = cpu_ref.mv_resize(flow, hp, wp)
        assert out.dtype == torch.float64
        assert maxdiff(out, g[f"out_{hp}x{wp}"]) <= 1e-12


@pytest.mark.parametrize("ci", range(5))
@pytest.mark.parametrize("seed", [0, 1])
def test_g3_my_attention(golden, ci, seed):
    g = golden(f"g3_attn_c{ci}_s{seed}")
    sd = {k[2:]: t(g[k]) for k in g.files if k.startswith("w.")}
    k = int(g["k"])
    out = cpu_ref.my_attention(sd, "", t(g["hr"]), t(g["lr"]), k, k)
    assert maxdiff(out, g["out"]) <= TOL


def test_g3_local_attention_pair(golden):
    g = golden("g3_pair")
    kH, kW = int(g["kH"]), int(g["kW"])
    v, w, q = t(g["v"]), t(g["w"]), t(g["q"])
    # weighting: pinned by the reference's in-tree f_weighting_cpu (attention.py:75-85)
    assert maxdiff(cpu_ref.local_weighting(v, w, kH, kW), g["weighting_cpu"]) <= TOL
    assert maxdiff(c_ref.local_weighting(g["v"], g["w"], kH, kW), g["weighting_cpu"]) <= TOL
    assert maxdiff(g["weighting_shim"], g["weighting_cpu"]) <= TOL
    # similar: contract from attention.py:56-64 comments (parity unpinned w.r.t. the absent CUDA op)
    assert maxdiff(cpu_ref.local_similar(q, v, kH, kW), g["similar_shim"]) <= TOL
    assert maxdiff(c_ref.local_similar(g["q"], g["v"], kH, kW), g["similar_shim"]) <= TOL


def test_c_oracle_mt_equals_scalar(golden):
    """oracle/local_attn_ref.c's OpenMP variants (what bench.py's CPU leg times on every host core) == its scalar functions bit for bit, on the
    reference-generated pair fixture and on a ragged random case; cpu_ref routed through them reproduces G3."""
    g = golden("g3_pair")
    kH, kW = int(g["kH"]), int(g["kW"])
    assert np.array_equal(c_ref.local_similar_mt(g["q"], g["v"], kH, kW, 3), c_ref.local_similar(g["q"], g["v"], kH, kW))
    assert np.array_equal(c_ref.local_weighting_mt(g["v"], g["w"], kH, kW, 3), c_ref.local_weighting(g["v"], g["w"], kH, kW))
    rng = np.random.default_rng(5)
    q, k = rng.standard_normal((2, 5, 9, 13), dtype=np.float32), rng.standard_normal((2, 5, 9, 13), dtype=np.float32)
    w = rng.standard_normal((2, 9, 13, 15), dtype=np.float32)
    assert np.array_equal(c_ref.local_similar_mt(q, k, 3, 5, 4), c_ref.local_similar(q, k, 3, 5))
    assert np.array_equal(c_ref.local_weighting_mt(q, w, 3, 5, 4), c_ref.local_weighting(q, w, 3, 5))
    g = golden("g3_attn_c1_s0")
    sd = {k_[2:]: t(g[k_]) for k_ in g.files if k_.startswith("w.")}
    kk = int(g["k"])
    prev = cpu_ref.use_c_local_attention(2)
    try:
        out = cpu_ref.my_attention(sd, "", t(g["hr"]), t(g["lr"]), kk, kk)
    finally:
        cpu_ref.use_c_local_attention(prev)
    assert maxdiff(out, g["out"]) <= TOL


def test_g4_pspnet(golden, manifest):
    g = golden("g4_pspnet")
    sd = sd_from_manifest(manifest, "PSPNet", 0)
    out, cls, p = cpu_ref.pspnet_forward(sd, t(g["x"]))
    assert maxdiff(p, g["p"]) <= 5e-5
    assert maxdiff(out, g["out"]) <= 5e-5
    assert maxdiff(cls, g["cls"]) <= 5e-5


def test_g5_pspnet_with_fuse(golden, manifest):
    g = golden("g5_pspfuse")
    ref_p = t(golden("g4_pspnet")["p"])
    sd = sd_from_manifest(manifest, "PSPNetWithFuse", 1)
    cls1, p1 = cpu_ref.pspnet_fuse_phase1(sd, t(g["x"]))
    assert maxdiff(cls1, g["cls1"]) <= 5e-5 and maxdiff(p1, g["p1"]) <= 5e-5
    out2, p2 = cpu_ref.pspnet_fuse_phase2(sd, t(g["p1"]), ref_p)
    assert maxdiff(out2, g["out2"]) <= 5e-5 and maxdiff(p2, g["p2"]) <= 5e-5
    outn, clsn, pn = cpu_ref.pspnet_forward(sd, t(g["x"]))
    assert maxdiff(outn, g["out_normal"]) <= 5e-5 and maxdiff(pn[..., ::2, ::2], g["p_normal_s2"]) <= 5e-5
    assert bool(g["merge_equal"].all())


def test_g8_pspnet_semseg(golden, manifest):
    """Cityscapes PSPNet-18 (model/pspnet_semseg.py): the same state_dict spec serves all three nets (seeds 4, 5, 6)."""
    g = golden("g8_semseg")
    sd_hr = sd_from_manifest(manifest, "SemsegPSPNetWithFuse", 4)
    out, aux, p = cpu_ref.semseg_forward(sd_hr, t(g["x"]))
    assert maxdiff(out, g["out"]) <= 5e-5 and maxdiff(aux, g["aux"]) <= 5e-5 and maxdiff(p, g["p"]) <= 5e-5
    sd_lr = sd_from_manifest(manifest, "SemsegPSPNetWithFuse", 5)
    x_tmp, p1 = cpu_ref.semseg_phase1(sd_lr, t(g["xl"]))
    assert maxdiff(x_tmp, g["x_tmp1"]) <= 5e-5 and maxdiff(p1, g["p1"]) <= 5e-5
    out2, p2 = cpu_ref.semseg_phase2(sd_lr, t(g["p1"]), t(g["p"]))
    assert maxdiff(out2, g["out2"]) <= 5e-5 and maxdiff(p2, g["p2"]) <= 5e-5
    assert maxdiff(cpu_ref.semseg_forward(sd_lr, t(g["xl"]))[1], g["aux_merge"]) <= 5e-5
    assert bool(g["merge_equal"].all())
    sd_plain = sd_from_manifest(manifest, "SemsegPSPNet", 6)
    assert maxdiff(cpu_ref.semseg_forward(sd_plain, t(g["x"]))[0], g["out_plain"]) <= 5e-5


def test_g6_bisenet(golden, manifest):
    g = golden("g6_bisenet")
    sd = sd_from_manifest(manifest, "BiSeNetV1", 2)
    out, o16, o32, fuse = cpu_ref.bisenet_forward(sd, t(g["x"]))
    assert maxdiff(out, g["out"]) <= 5e-5
    assert maxdiff(o16[..., ::4, ::4], g["out16_s4"]) <= 5e-5
    assert maxdiff(o32[..., ::4, ::4], g["out32_s4"]) <= 5e-5
    assert maxdiff(fuse, g["feat_fuse"]) <= 5e-5
    # odd sizes: feat8 (9x17) != 2*feat16 (10x18) -> the re-interpolation branches (bisenet.py:298,442)
    go = golden("g6_biseodd")
    oo, _, _, fo = cpu_ref.bisenet_forward(sd, t(go["x"]))
    assert maxdiff(oo[..., ::2, ::2], go["hr_out_s2"]) <= 5e-5 and maxdiff(fo, go["hr_feat_fuse"]) <= 5e-5


def test_g6_bisenet_with_fuse(golden, manifest):
    g = golden("g6_bisefuse")
    sd = sd_from_manifest(manifest, "BiSeNetV1WithFuse", 3)
    a16, a32, mid = cpu_ref.bisenet_fuse_phase1(sd, t(g["x"]))
    assert maxdiff(mid, g["mid"]) <= 5e-5
    assert maxdiff(a16[..., ::4, ::4], g["aux16_s4"]) <= 5e-5 and maxdiff(a32[..., ::4, ::4], g["aux32_s4"]) <= 5e-5
    out, p = cpu_ref.bisenet_fuse_phase2(sd, t(g["mid"]), t(g["ref_p"]))
    assert maxdiff(out, g["out"]) <= 5e-5 and maxdiff(p, g["p"]) <= 5e-5
    go = golden("g6_biseodd")
    a16o, _, mido = cpu_ref.bisenet_fuse_phase1(sd, t(go["x"]))
    assert maxdiff(mido, go["mid"]) <= 5e-5 and maxdiff(a16o[..., ::4, ::4], go["aux16_s4"]) <= 5e-5


@pytest.mark.parametrize("kind,hr_name,hr_seed,lr_name,lr_seed", [("psp", "PSPNet", 0, "PSPNetWithFuse", 1),
                                                                   ("bise", "BiSeNetV1", 2, "BiSeNetV1WithFuse", 3)])
def test_g7_alter_res_step(golden, manifest, kind, hr_name, hr_seed, lr_name, lr_seed):
    g = golden(f"g7_alter_{kind}")
    sd_hr = sd_from_manifest(manifest, hr_name, hr_seed)
    sd_lr = sd_from_manifest(manifest, lr_name, lr_seed)
    flow = cpu_ref.mv_from_int16(t(g["mvq"]))
    out, p, warped, _ = cpu_ref.alter_res_step(kind, sd_hr, sd_lr, t(g["img"]), t(g["ref"]), flow, 0.5)
    assert maxdiff(warped, g["warped"]) <= 5e-5
    assert maxdiff(out, g["out"]) <= 1e-4
    assert maxdiff(p[..., ::2, ::2], g["p_s2"]) <= 1e-4
    preds, hist = cpu_ref.eval_tail(out, t(g["label"]), 12)
    assert (preds.numpy() != g["preds"]).mean() <= 1e-3          # argmax ties at 1e-5 noise only
    assert abs(float(cpu_ref.miou(t(g["hist"]))) - float(g["miou"])) <= 1e-6
    assert float((hist - t(g["hist"])).abs().sum()) <= 4


@pytest.mark.parametrize("kind,hr_name,lr_name", [("psp", "PSPNet", "PSPNetWithFuse"), ("bise", "BiSeNetV1", "BiSeNetV1WithFuse")])
def test_g10_undamped_alter_res_step(golden, manifest, kind, hr_name, lr_name):
    """The oracle against the reference on UN-DAMPED weights (plain He initialisation: activations reach 200-700, the CReFF softmax
    is sharp).  Tolerances are relative to the tensor's own magnitude: two fp32 CPU evaluations of the same network already
    differ by ~1e-6 of it."""
    g = golden(f"g10_undamped_{kind}")
    sd_hr = sd_from_manifest(manifest, hr_name, 20, attn_gain=1.0, res_gain=1.0)
    sd_lr = sd_from_manifest(manifest, lr_name, 21, attn_gain=1.0, res_gain=1.0)
    out, p, warped, _ = cpu_ref.alter_res_step(kind, sd_hr, sd_lr, t(g["img"]), t(g["ref"]), cpu_ref.mv_from_int16(t(g["mvq"])), 0.5)
    sub = 2 if kind == "psp" else 1
    a_w, a_lr, a_p, a_o = (float(v) for v in g["abs_max"])
    assert maxdiff(warped[..., ::sub, ::sub], g["warped_s"]) <= 2e-6 * a_w
    assert maxdiff(p[..., ::sub, ::sub], g["p_s"]) <= 2e-6 * a_p
    assert maxdiff(out, g["out"]) <= 2e-6 * a_o
    preds, _ = cpu_ref.eval_tail(out, t(g["label"]), 12)
    assert (preds.numpy() != g["preds"]).mean() <= 1e-3


def test_g9_merge_motion(golden):
    """mergeMotion restatement vs the reference function's own output on the same seeded 720x960 motion fields (G9):
    strided sample, a dense crop of the last frame and the SHA-256 of the full int32 array."""
    import hashlib

    from arseg_amd import synth

    g = golden("g9_mergemotion")
    F_ = int(g["F"])
    flows = synth.make_mv_chain(int(g["seed"]), 720, 960, F_)
    out = cpu_ref.merge_motion(flows)
    assert out.shape == (720, 960, F_ + 1, 2) and out.dtype == np.int32
    assert np.array_equal(out[::9, ::8], g["out_s"]) and np.array_equal(out[100:164, 200:296, F_], g["frame_last_crop"])
    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest() == g["sha256"].tobytes()
