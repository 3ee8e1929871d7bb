"""GPU parity tests, model level: the nn.Module mirrors (same interface as the reference's model/*.py) against the
golden vectors produced by the reference itself and against the oracle.  Tolerance: 1e-3 abs in fp32 (north star)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import maxdiff, sd_from_manifest, t

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from arseg_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _psp(manifest, dev, fuse, seed=None, gains=(0.12, 0.3)):
    from arseg_amd.model import PSPNet, PSPNetWithFuse

    if fuse:
        m = PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18", atten_k=7)
        name, dseed = "PSPNetWithFuse", 1
    else:
        m = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
        name, dseed = "PSPNet", 0
    seed = dseed if seed is None else seed
    from arseg_amd import synth

    spec = [(k, tuple(s)) for k, s in manifest[name]["keys"]]
    sd = {"module." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, seed, *gains).items()}   # DataParallel-style keys
    m = torch.nn.DataParallel(m)
    m.load_state_dict(sd)                                                                              # evaluation.py:41-46
    return m.module.to(dev).eval()


def _bise(manifest, dev, fuse, seed=None, gains=(0.12, 0.3)):
    from arseg_amd import synth
    from arseg_amd.model import BiSeNetV1, BiSeNetV1WithFuse

    m = BiSeNetV1WithFuse(n_classes=12, backend="resnet18") if fuse else BiSeNetV1(n_classes=12, backend="resnet18")
    name, dseed = ("BiSeNetV1WithFuse", 3) if fuse else ("BiSeNetV1", 2)
    seed = dseed if seed is None else seed
    spec = [(k, tuple(s)) for k, s in manifest[name]["keys"]]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, seed, *gains).items()})
    return m.to(dev).eval()


def test_pspnet_golden(dev, golden, manifest):
    g = golden("g4_pspnet")
    net = _psp(manifest, dev, False)
    with torch.no_grad():
        out, cls, p = net(t(g["x"]).to(dev))
    assert out.shape == g["out"].shape and cls.shape == g["cls"].shape and p.shape == g["p"].shape
    assert maxdiff(p, g["p"]) <= TOL and maxdiff(out, g["out"]) <= TOL and maxdiff(cls, g["cls"]) <= TOL


def test_pspnet_with_fuse_golden(dev, golden, manifest):
    g = golden("g5_pspfuse")
    ref_p = t(golden("g4_pspnet")["p"]).to(dev)
    net = _psp(manifest, dev, True)
    with torch.no_grad():
        cls1, p1 = net.forward_phase1(t(g["x"]).to(dev))
        assert maxdiff(cls1, g["cls1"]) <= TOL and maxdiff(p1, g["p1"]) <= TOL
        out2, p2 = net.forward_phase2(t(g["p1"]).to(dev), ref_p)                               # NCHW-contiguous inputs
        assert maxdiff(out2, g["out2"]) <= TOL and maxdiff(p2, g["p2"]) <= TOL
        out2b, p2b = net.forward_phase2(p1, ref_p.contiguous(memory_format=torch.channels_last))  # channels_last inputs
        assert maxdiff(out2b, g["out2"]) <= TOL and maxdiff(p2b, g["p2"]) <= TOL
        outn, clsn, pn = net(t(g["x"]).to(dev), mode="normal")
        assert maxdiff(outn, g["out_normal"]) <= TOL and maxdiff(pn[..., ::2, ::2], g["p_normal_s2"]) <= TOL
        outm, clsm, pm = net(t(g["x"]).to(dev), mode="merge", ref_p=ref_p)
        assert maxdiff(outm, g["out2"]) <= TOL and maxdiff(clsm, g["cls1"]) <= TOL


def _semseg(manifest, dev, seed, fuse=True):
    from arseg_amd import synth
    from arseg_amd.model import pspnet_semseg

    cls = pspnet_semseg.PSPNetWithFuse if fuse else pspnet_semseg.PSPNet
    name = "SemsegPSPNetWithFuse" if fuse else "SemsegPSPNet"
    m = cls(bins=(1, 2, 3, 6), classes=19, feat_dim=512, layers=18)
    spec = [(k, tuple(s)) for k, s in manifest[name]["keys"]]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, seed).items()})
    return m.to(dev).eval()


def test_pspnet_semseg_golden(dev, golden, manifest):
    """Cityscapes PSPNet-18: HR branch, phase 1, phase 2 (CReFF at C=512 on the matrix-core kernel) and
    the 'merge' mode against vectors produced by the reference's model/pspnet_semseg.py."""
    g = golden("g8_semseg")
    hr, lr = _semseg(manifest, dev, 4), _semseg(manifest, dev, 5)
    with torch.no_grad():
        out, aux, p = hr(t(g["x"]).to(dev))
        assert out.shape == g["out"].shape and aux.shape == g["aux"].shape and p.shape == g["p"].shape
        assert maxdiff(out, g["out"]) <= TOL and maxdiff(aux, g["aux"]) <= TOL and maxdiff(p, g["p"]) <= TOL
        x_tmp, p1 = lr.forward_phase1(t(g["xl"]).to(dev))
        assert maxdiff(x_tmp, g["x_tmp1"]) <= TOL and maxdiff(p1, g["p1"]) <= TOL
        out2, p2 = lr.forward_phase2(t(g["p1"]).to(dev), t(g["p"]).to(dev))
        assert out2.shape == g["out2"].shape
        assert maxdiff(out2, g["out2"]) <= TOL and maxdiff(p2, g["p2"]) <= TOL
        outm, auxm, pm = lr(t(g["xl"]).to(dev), mode="merge", ref_p=p)                 # ref_p as the HR net returned it
        assert maxdiff(outm, g["out2"]) <= TOL and maxdiff(auxm, g["aux_merge"]) <= TOL and maxdiff(pm, g["p2"]) <= TOL
        (outp,) = _semseg(manifest, dev, 6, fuse=False)(t(g["x"]).to(dev))
        assert maxdiff(outp, g["out_plain"]) <= TOL


def test_bisenet_golden(dev, golden, manifest):
    g = golden("g6_bisenet")
    net = _bise(manifest, dev, False)
    with torch.no_grad():
        out, o16, o32, fuse = net(t(g["x"]).to(dev))
    assert maxdiff(out, g["out"]) <= TOL and maxdiff(fuse, g["feat_fuse"]) <= TOL
    assert maxdiff(o16[..., ::4, ::4], g["out16_s4"]) <= TOL and maxdiff(o32[..., ::4, ::4], g["out32_s4"]) <= TOL
    go = golden("g6_biseodd")                                            # odd sizes: the re-interpolation branches
    with torch.no_grad():
        oo, _, _, fo = net(t(go["x"]).to(dev))
    assert maxdiff(oo[..., ::2, ::2], go["hr_out_s2"]) <= TOL and maxdiff(fo, go["hr_feat_fuse"]) <= TOL


def test_bisenet_with_fuse_golden(dev, golden, manifest):
    g = golden("g6_bisefuse")
    net = _bise(manifest, dev, True)
    with torch.no_grad():
        a16, a32, mid = net.forward_phase1(t(g["x"]).to(dev))
        assert maxdiff(mid, g["mid"]) <= TOL
        assert maxdiff(a16[..., ::4, ::4], g["aux16_s4"]) <= TOL and maxdiff(a32[..., ::4, ::4], g["aux32_s4"]) <= TOL
        out, p = net.forward_phase2(t(g["mid"]).to(dev), t(g["ref_p"]).to(dev))
        assert maxdiff(out, g["out"]) <= TOL and maxdiff(p, g["p"]) <= TOL
        go = golden("g6_biseodd")
        a16o, _, mido = net.forward_phase1(t(go["x"]).to(dev))
        assert maxdiff(mido, go["mid"]) <= TOL and maxdiff(a16o[..., ::4, ::4], go["aux16_s4"]) <= TOL


@pytest.mark.parametrize("kind", ["psp", "bise"])
def test_eval_alter_res_golden(dev, golden, manifest, kind):
    """The reference's EvalAlterRes step (evaluation.py:161-209) through the drop-in interface and through the fast path."""
    from arseg_amd import evaluation as ev
    from arseg_amd import _lib, ops

    g = golden(f"g7_alter_{kind}")
    hr = (_psp if kind == "psp" else _bise)(manifest, dev, False)
    lr = (_psp if kind == "psp" else _bise)(manifest, dev, True)
    img, ref, label = t(g["img"]), t(g["ref"]), t(g["label"])
    flow = t(g["mvq"]).double() / 4
    captured = {}
    orig = lr.forward_phase2

    def spy(p, ref_p):
        r = orig(p, ref_p)
        captured["warped"], captured["out"], captured["p"] = ref_p, r[0], r[1]
        return r

    lr.forward_phase2 = spy
    with torch.no_grad():
        miou = ev.EvalAlterRes(scale=0.5)(hr, torch.nn.DataParallel(lr), [(img, label, None, ref, flow)], 12)
    lr.forward_phase2 = orig
    assert maxdiff(captured["warped"], g["warped"]) <= TOL
    assert maxdiff(captured["out"], g["out"]) <= TOL
    assert maxdiff(captured["p"][..., ::2, ::2], g["p_s2"]) <= TOL
    assert abs(miou - float(g["miou"])) <= 2e-3
    with torch.no_grad():
        ref_p = hr(ref.to(dev))[-1]
        out_f, p_c8 = ev.alter_res_step_fast(lr, ops.to_nhwc(ref_p), img.to(dev), t(g["mvq"]).to(dev), 0.5)
        pred, hist = ops.argmax_confusion(out_f, label.to(dev), label.shape[-2], label.shape[-1])
    assert maxdiff(out_f, g["out"]) <= TOL
    assert maxdiff(ops.from_c8(p_c8, _lib.NCHW)[..., ::2, ::2], g["p_s2"]) <= TOL
    # index outputs: the labels are EXACTLY the reference's wherever the reference's top-2 margin exceeds twice the
    # measured logit error -- no argmax can flip there; the remaining near-ties are counted, bounded and the only source of histogram moves
    err = maxdiff(out_f, g["out"])
    top2g = torch.from_numpy(np.asarray(g["out"])).topk(2, dim=1).values
    safe = ((top2g[:, 0] - top2g[:, 1]) > 2 * err + 1e-7).numpy()
    n_tie = int((~safe).sum())
    print(f"\n[g7 {kind}] logit err {err:.2e}; labels compared exactly on {int(safe.sum())} of {safe.size} pixels, {n_tie} near-ties excluded")
    assert np.array_equal(pred.cpu().long().numpy()[safe], g["preds"][safe])
    assert n_tie <= 2e-3 * safe.size
    assert float((hist.cpu().float() - t(g["hist"])).abs().sum()) <= 2 * n_tie
    # fused evaluator tail (BiSeNet: head -> x8 upsample -> argmax without the full-resolution logits) against the reference's
    # preds / confusion matrix: labels may differ only where the reference's top two classes are within 1e-4 of each other
    with torch.no_grad():
        pred2, hist2 = ev.alter_res_batch_pred(lr, [ops.to_nhwc(ref_p)[0]], img.to(dev), t(g["mvq"]).to(dev), 0.5, labels=label.to(dev))
    top2 = torch.from_numpy(np.asarray(g["out"])).topk(2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 1e-4).numpy()
    assert np.array_equal(pred2.cpu().long().numpy()[clear], g["preds"][clear])
    assert float((hist2.cpu().float() - t(g["hist"])).abs().sum()) <= 2 * int((~clear).sum())
    with torch.no_grad():
        miou_c = ev.EvalConstRes(scale=1.0)(hr, [(ref, label, None)], 12)
    assert abs(miou_c - manifest[f"g7_{kind}_miou_const"]) <= 2e-3


@pytest.mark.parametrize("kind", ["psp", "bise"])
def test_eval_alter_res_undamped_golden(dev, golden, manifest, kind):
    """Parity away from the conditioned weights: the reference's EvalAlterRes step with plain He initialisation
    everywhere (G10: attn_gain = res_gain = 1.0).  Activations reach 100-700 and the CReFF scores q.k reach 1e4-1e5, i.e. one fp32
    ulp of a score is ~5e-3: the softmax amplifies rounding, and ANY fp32 evaluation -- the reference's own included -- is only
    defined to ~1e-2 absolute here.  What is asserted:
      * backbone outputs (keyframe feature, LR feature): |err| <= 1e-5 x magnitude  (measured 4e-6);
      * the CReFF stage on identical inputs, against an fp64 evaluation of the same formula: the HIP kernel is no further from it
        than 6x the reference-order fp32 CPU evaluation is (measured 3.4x: the hi/lo fp16 split carries 22 of fp32's 24 significand
        bits, i.e. 4x the operand rounding; the fp32 VALU kernel measures 2x);
      * end to end: |err| <= 1e-4 x magnitude and >= 99.8 % identical labels.
    The 1e-3 absolute bound of the north star is met at the O(1-30) activations of G4-G8; it is NOT meaningful at this conditioning
    (the measured figures are printed with pytest -s and quoted in DESIGN.md)."""
    from arseg_amd import evaluation as ev
    from arseg_amd import _lib, ops
    from oracle import cpu_ref

    g = golden(f"g10_undamped_{kind}")
    hr = (_psp if kind == "psp" else _bise)(manifest, dev, False, seed=20, gains=(1.0, 1.0))
    lr = (_psp if kind == "psp" else _bise)(manifest, dev, True, seed=21, gains=(1.0, 1.0))
    img, ref, label, mvq = t(g["img"]), t(g["ref"]), t(g["label"]), t(g["mvq"])
    sub = 2 if kind == "psp" else 1
    a_w, a_lr, a_p, a_o = (float(v) for v in g["abs_max"])
    sd_hr = sd_from_manifest(manifest, "PSPNet" if kind == "psp" else "BiSeNetV1", 20, 1.0, 1.0)
    sd_lr = sd_from_manifest(manifest, "PSPNetWithFuse" if kind == "psp" else "BiSeNetV1WithFuse", 21, 1.0, 1.0)
    with torch.no_grad():
        o_out, o_p, o_warp, o_ref = cpu_ref.alter_res_step(kind, sd_hr, sd_lr, img, ref, cpu_ref.mv_from_int16(mvq), 0.5)
        ref_p = hr(ref.to(dev))[-1]
        out_f, p_c8 = ev.alter_res_step_fast(lr, ops.to_nhwc(ref_p), img.to(dev), mvq.to(dev), 0.5)
        pred, _ = ops.argmax_confusion(out_f, label.to(dev), label.shape[-2], label.shape[-1])
        # CReFF stage alone, identical (oracle) inputs, against fp64
        lr_p = t(g["lr_p_s"]) if sub == 1 else None
        h, w = img.shape[-2] // 2, img.shape[-1] // 2
        o_lr = (cpu_ref.pspnet_fuse_phase1 if kind == "psp" else cpu_ref.bisenet_fuse_phase1)(
            sd_lr, F.interpolate(img, (h, w), mode="bilinear", align_corners=True))[-1]
        pre = "fuse_attention."
        sd64 = {k: v.double() for k, v in sd_lr.items() if k.startswith(pre)}
        p64 = cpu_ref.my_attention(sd64, pre, o_warp.double(), o_lr.double(), 7, 7)
        p32 = cpu_ref.my_attention(sd_lr, pre, o_warp, o_lr, 7, 7)
        _, pg = lr.phase2_warp(ops.to_nhwc(o_lr.to(dev)).contiguous(), [ops.to_nhwc(o_ref.to(dev))[0].contiguous()], mvq.to(dev))
    e_ref = maxdiff(ref_p, o_ref)
    e_stage_gpu, e_stage_cpu32 = maxdiff(ops.from_c8(pg, _lib.NCHW), p64), maxdiff(p32, p64)
    e_out, e_p = maxdiff(out_f, g["out"]), maxdiff(ops.from_c8(p_c8, _lib.NCHW)[..., ::sub, ::sub], g["p_s"])
    agree = float((pred.cpu().long().numpy() == g["preds"]).mean())
    print(f"\n[undamped {kind}] keyframe feature err {e_ref:.2e} (max {a_w:.0f}); CReFF stage vs fp64: HIP {e_stage_gpu:.2e}, fp32 CPU {e_stage_cpu32:.2e} "
          f"(max {a_p:.0f}); end to end: logits {e_out:.2e} (max {a_o:.0f}), p {e_p:.2e}, labels equal {agree:.5f}")
    assert e_ref <= 1e-5 * a_w
    assert e_stage_gpu <= 6 * e_stage_cpu32 + 1e-6 * a_p
    # end to end the sharp softmax amplifies the fp32 rounding of the backbone: measured 0.8e-4 .. 2.1e-4 of the tensor's magnitude depending
    # on which conv plans the autotuner picked (Winograd / direct / split-K change the summation order); the fp32 CPU restatement of the
    # CReFF stage alone sits 1.5e-5 of the magnitude away from fp64 (printed above)
    assert e_p <= 4e-4 * a_p and e_out <= 4e-4 * a_o
    assert agree >= 0.998
    # index outputs: exact on every pixel whose reference top-2 margin exceeds twice the measured logit error
    top2g = torch.from_numpy(np.asarray(g["out"])).topk(2, dim=1).values
    safe = ((top2g[:, 0] - top2g[:, 1]) > 2 * e_out + 1e-7).numpy()
    print(f"[undamped {kind}] labels compared exactly on {int(safe.sum())} of {safe.size} pixels ({int((~safe).sum())} within 2 x the logit error of a tie)")
    assert np.array_equal(pred.cpu().long().numpy()[safe], g["preds"][safe])


def test_evaluator_range_fallback(dev, golden, manifest, monkeypatch):
    """An evaluation pass whose activations leave the split-fp16 operand range (a frame scaled by 3e5) is repeated on the fp32 back end:
    the mIoU equals the one computed under set_conv_math("f32") from the start, and an in-range pass is not repeated."""
    from arseg_amd import evaluation as ev
    from arseg_amd import ops

    g = golden("g7_alter_psp")
    hr = _psp(manifest, dev, False)
    img, label = t(g["ref"]), t(g["label"])
    calls = []
    orig = ops.set_conv_math
    monkeypatch.setattr(ops, "set_conv_math", lambda name: (calls.append(name), orig(name))[1])
    with torch.no_grad():
        ev.EvalConstRes(scale=1.0)(hr, [(img, label, None)], 12)
        assert calls == []                                       # in range: one pass
        m_big = ev.EvalConstRes(scale=1.0)(hr, [(img * 3e5, label, None)], 12)
        assert calls and calls[0] == "f32"                       # the device word was set: repeated on the fp32 MFMA
        prev = orig("f32")
        try:
            m_f32 = ev.EvalConstRes(scale=1.0)(hr, [(img * 3e5, label, None)], 12)
        finally:
            orig(prev)
        with pytest.raises(Exception):
            ev.EvalConstRes(scale=1.0)(hr, iter([(img * 3e5, label, None)]), 12)      # a one-shot iterator cannot be replayed
    assert m_big == m_f32 or (m_big != m_big and m_f32 != m_f32)


def test_modules_fail_loudly_off_gpu(manifest):
    """No CPU fallback: a forward on CPU tensors / CPU parameters raises instead of silently computing elsewhere."""
    from arseg_amd import _lib
    from arseg_amd.model import MyAttention

    m = MyAttention(8, kW=7, kH=7).eval()
    with pytest.raises(_lib.ArsegError):
        m(torch.zeros(1, 8, 8, 8), torch.zeros(1, 8, 4, 4))


@pytest.mark.parametrize("kind,H,W", [("psp", 512, 1024), ("bise", 1024, 2048)])
def test_full_size_properties(dev, kind, H, W):
    """BASELINE.json sizes, where the CPU oracle is too slow to run whole: size-independent properties of the path.
    (1) determinism: two runs are bit-identical; (2) log-probabilities normalise (PSPNet); (3) with zero q/k/v weights the
    CReFF output is lr_up + V-bias * (in-image tap fraction): interior pixels get exactly the bias; (4) a strip of the
    output far from the strip's edges equals the same computation on the cropped inputs (locality: 7x7 window + 3x3 convs)."""
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention

    C, Hp, Wp, hp, wp, n_cls = (64, H, W, H // 2, W // 2, 12) if kind == "psp" else (256, H // 8, W // 8, H // 16, W // 16, 19)
    g = torch.Generator(device="cpu").manual_seed(5)
    hr = torch.randn(1, Hp, Wp, C, generator=g).to(dev)
    lr = torch.randn(1, hp, wp, C, generator=g).to(dev)
    m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 3)
    pa = PackedAttention(m, dev)
    wf, bf = (0.2 * torch.randn(n_cls, C, generator=g)).to(dev), (0.1 * torch.randn(n_cls, generator=g)).to(dev)
    hr_c8 = ops.to_c8(hr, _lib.NHWC)
    p1, l1 = ops.creff(hr_c8, lr, pa, (wf, bf), kind == "psp")
    p2, l2 = ops.creff(hr_c8, lr, pa, (wf, bf), kind == "psp")
    assert torch.equal(p1, p2) and torch.equal(l1, l2)
    assert torch.isfinite(p1).all() and torch.isfinite(l1).all()
    if kind == "psp":
        assert float((l1.exp().sum(dim=1) - 1).abs().max()) <= 1e-4
    # locality: rows [r0, r1) of the output depend on hr rows [r0-4, r1+4) and the matching lr rows only through lr_up,
    # which we feed directly (lr at full resolution => the upsample is the identity)
    lr_full = torch.randn(1, Hp, Wp, C, generator=g).to(dev)
    pf, _ = ops.creff(hr_c8, lr_full, pa, None, False)
    r0, r1 = Hp // 2 - 8, Hp // 2 + 8
    crop_hr = ops.to_c8(hr[:, r0 - 8:r1 + 8].contiguous(), _lib.NHWC)
    pc, _ = ops.creff(crop_hr, lr_full[:, r0 - 8:r1 + 8].contiguous(), pa, None, False)
    assert maxdiff(ops.from_c8(pf, _lib.NHWC)[:, r0:r1], ops.from_c8(pc, _lib.NHWC)[:, 8:8 + (r1 - r0)]) <= 1e-5


def test_full_size_end_to_end_headline(dev):
    """BASELINE configs[1] at full size inside the suite: PSPNet-18 keyframe HR forward at 512x1024, one non-keyframe through
    downscale -> LR backbone (256x512) -> MV warp + CReFF + head (fused kernel) against the CPU oracle, 1e-3 abs (north star);
    the fused kernel also against the two-kernel path.  ~10 s of CPU oracle on the box's host cores."""
    from arseg_amd import evaluation as ev
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import PSPNet, PSPNetWithFuse
    from oracle import cpu_ref

    H, W = 512, 1024
    hr = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
    lr = PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18", atten_k=7)
    synth.load_synth_weights(hr, 0)
    synth.load_synth_weights(lr, 1)
    sd_hr = synth.resolve_aliases({k: v.clone() for k, v in hr.state_dict().items()})
    sd_lr = synth.resolve_aliases({k: v.clone() for k, v in lr.state_dict().items()})
    hr, lr = hr.to(dev).eval(), lr.to(dev).eval()
    clip = synth.make_clip(2, H, W, gop=6, mean=synth.CAMVID_MEAN, std=synth.CAMVID_STD)
    key, img, mvq = (torch.from_numpy(clip[k][i:i + 1]) for k, i in (("frames", 0), ("frames", 5), ("mv", 5)))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        o_out, o_p, _, o_ref = cpu_ref.alter_res_step("psp", sd_hr, sd_lr, img, key, cpu_ref.mv_from_int16(mvq), 0.5)
        ref_p = hr(key.to(dev))[-1]
        out, p_c8 = ev.alter_res_step_fast(lr, ops.to_nhwc(ref_p), img.to(dev), mvq.to(dev), 0.5)
        feat = lr.phase1_nhwc4(ops.frame_to_nhwc4(img.to(dev), H // 2, W // 2))[-1]
        out2, p2 = lr.phase2_c8(feat, ops.warp_mvq(ops.to_nhwc(ref_p).contiguous(), mvq.to(dev), _lib.C8))      # two-kernel path
    assert maxdiff(ref_p, o_ref) <= 1e-3
    assert maxdiff(out, o_out) <= 1e-3 and maxdiff(ops.from_c8(p_c8, _lib.NCHW), o_p) <= 1e-3
    assert float((out.argmax(1).cpu() == o_out.argmax(1)).float().mean()) >= 0.9999
    assert maxdiff(out, out2) <= 2e-4 and maxdiff(p_c8, p2) <= 2e-4


def test_full_size_camvid_ar_step(dev):
    """The reference's own dataset size: CamVid 720x960, AR-Seg 0.5x (evaluation.py --mode 1 1 1 on camvid-psp18): keyframe HR forward,
    one non-keyframe through downscale (360x480) -> LR backbone -> MV warp + CReFF + head, HIP against the CPU oracle, 1e-3 abs, and the
    labels exactly wherever the oracle's top-2 margin exceeds twice the measured logit error.  The 720x960 map is 45 x 60 tiles of the fused
    kernel; the LR map 360x480 gives odd 45x60 stride-8 maps in the backbone."""
    from arseg_amd import evaluation as ev
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import PSPNet, PSPNetWithFuse
    from oracle import cpu_ref

    H, W = 720, 960
    hr = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
    lr = PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18", atten_k=7)
    synth.load_synth_weights(hr, 4)
    synth.load_synth_weights(lr, 5)
    sd_hr = synth.resolve_aliases({k: v.clone() for k, v in hr.state_dict().items()})
    sd_lr = synth.resolve_aliases({k: v.clone() for k, v in lr.state_dict().items()})
    hr, lr = hr.to(dev).eval(), lr.to(dev).eval()
    clip = synth.make_clip(7, H, W, gop=12, mean=synth.CAMVID_MEAN, std=synth.CAMVID_STD)
    key, img, mvq = (torch.from_numpy(clip[k][i:i + 1]) for k, i in (("frames", 0), ("frames", 11), ("mv", 11)))      # the farthest frame of the GOP
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        o_out, o_p, _, o_ref = cpu_ref.alter_res_step("psp", sd_hr, sd_lr, img, key, cpu_ref.mv_from_int16(mvq), 0.5)
        ref_p = hr(key.to(dev))[-1]
        out, p_c8 = ev.alter_res_step_fast(lr, ops.to_nhwc(ref_p), img.to(dev), mvq.to(dev), 0.5)
    e_out = maxdiff(out, o_out)
    assert maxdiff(ref_p, o_ref) <= 1e-3 and e_out <= 1e-3 and maxdiff(ops.from_c8(p_c8, _lib.NCHW), o_p) <= 1e-3
    top2 = o_out.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * e_out + 1e-7
    assert torch.equal(out.argmax(1).cpu()[safe], o_out.argmax(1)[safe])
    assert int((~safe).sum()) <= 2e-3 * safe.numel()


def test_full_size_bisenet_step(dev):
    """BASELINE configs[2] shapes in fp32: BiSeNet-18 keyframe HR forward at 1024x2048, one non-keyframe through downscale (512x1024) ->
    LR backbone -> MV resize + warp + CReFF (C = 256 at 128x256) + head + x8 upsample, HIP against the CPU oracle, 1e-3 abs; and the fused
    evaluator tail (argmax without the full-resolution logits) exactly on the pixels whose margin exceeds twice the logit error."""
    from arseg_amd import evaluation as ev
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import BiSeNetV1, BiSeNetV1WithFuse
    from oracle import cpu_ref

    H, W = 1024, 2048
    hr, lr = BiSeNetV1(n_classes=19, backend="resnet18"), BiSeNetV1WithFuse(n_classes=19, backend="resnet18")
    synth.load_synth_weights(hr, 6)
    synth.load_synth_weights(lr, 7)
    sd_hr = synth.resolve_aliases({k: v.clone() for k, v in hr.state_dict().items()})
    sd_lr = synth.resolve_aliases({k: v.clone() for k, v in lr.state_dict().items()})
    hr, lr = hr.to(dev).eval(), lr.to(dev).eval()
    clip = synth.make_clip(3, H, W, gop=4, mean=synth.CITY_BISE_MEAN, std=synth.CITY_BISE_STD)
    key, img, mvq = (torch.from_numpy(clip[k][i:i + 1]) for k, i in (("frames", 0), ("frames", 3), ("mv", 3)))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        o_out, o_p, _, o_ref = cpu_ref.alter_res_step("bise", sd_hr, sd_lr, img, key, cpu_ref.mv_from_int16(mvq), 0.5)
        ref_p = hr(key.to(dev))[-1]
        out, p_c8 = ev.alter_res_step_fast(lr, ops.to_nhwc(ref_p), img.to(dev), mvq.to(dev), 0.5)
        pred, _ = ev.alter_res_batch_pred(lr, [ops.to_nhwc(ref_p)[0]], img.to(dev), mvq.to(dev), 0.5)
    e_out = maxdiff(out, o_out)
    assert maxdiff(ref_p, o_ref) <= 1e-3 and e_out <= 1e-3 and maxdiff(ops.from_c8(p_c8, _lib.NCHW), o_p) <= 1e-3
    top2 = o_out.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * e_out + 1e-6
    assert torch.equal(pred.cpu().long()[safe], o_out.argmax(1)[safe])
    assert int((~safe).sum()) <= 2e-3 * safe.numel()


def test_full_size_hr_720x960(dev):
    """BASELINE configs[0] shape: PSPNet-18 HR branch on one 720x960 (CamVid) frame, HIP against the CPU oracle, 1e-3 abs."""
    from arseg_amd import synth
    from arseg_amd.model import PSPNet
    from oracle import cpu_ref

    hr = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
    synth.load_synth_weights(hr, 0)
    sd_hr = synth.resolve_aliases({k: v.clone() for k, v in hr.state_dict().items()})
    hr = hr.to(dev).eval()
    x = torch.from_numpy(synth.make_clip(4, 720, 960, gop=1, mean=synth.CAMVID_MEAN, std=synth.CAMVID_STD)["frames"][0:1])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        want = cpu_ref.pspnet_forward(sd_hr, x)
        got = hr(x.to(dev))
    assert len(got) == 3 and got[0].shape == (1, 12, 720, 960) and got[2].shape == (1, 64, 720, 960)
    for a, b in zip(got, want):
        assert maxdiff(a, b) <= 1e-3


@pytest.mark.parametrize("kind", ["psp", "bise"])
def test_batched_fast_path_equals_per_frame(dev, manifest, kind):
    """The batched GOP path (all non-keyframes of a GOP in one pass, evaluation.alter_res_batch_fast) computes exactly what
    the frame-by-frame fast path computes: frames are independent (evaluation.py:161-193)."""
    from arseg_amd import evaluation as ev
    from arseg_amd import ops, synth

    hr = (_psp if kind == "psp" else _bise)(manifest, dev, False)
    lr = (_psp if kind == "psp" else _bise)(manifest, dev, True)
    H, W = (64, 96) if kind == "psp" else (128, 256)
    clip = synth.make_clip(5, H, W, gop=4)
    frames = torch.from_numpy(clip["frames"]).to(dev)
    mvs = torch.from_numpy(clip["mv"]).to(dev)
    with torch.no_grad():
        ref_p = ops.to_nhwc(hr(frames[0:1])[-1])[0]
        out_b, p_b = ev.alter_res_batch_fast(lr, [ref_p] * 3, frames[1:4], mvs[1:4], 0.5)
        for i in range(3):
            out_i, p_i = ev.alter_res_step_fast(lr, ref_p.unsqueeze(0), frames[1 + i:2 + i], mvs[1 + i:2 + i], 0.5)
            # same kernels, same per-frame arithmetic; only the conv launch plans (tile / split-K / Winograd) may differ with M
            assert maxdiff(out_b[i:i + 1], out_i) <= 2e-4
            assert maxdiff(p_b[i:i + 1], p_i) <= 2e-4


@pytest.mark.parametrize("kind", ["psp", "bise", "semseg"])
def test_fast_paths_without_aux_outputs(dev, manifest, kind):
    """The build's fast paths skip the training-only auxiliary outputs the evaluator discards (evaluation.py:173-174 takes ``[-1]`` of the
    keyframe forward, :190-191 ``[-1]`` of forward_phase1): ``forward_keyframe`` / ``phase1_nhwc4(aux=False)`` must return exactly what
    ``forward`` / ``forward_phase1`` return at those positions (same launches, same plans), and ``forward`` itself still returns every output
    of the reference; ``ops.config.aux_outputs`` switches the fast paths back."""
    from arseg_amd import evaluation as ev
    from arseg_amd import ops, synth

    mk = {"psp": _psp, "bise": _bise}.get(kind)
    hr = mk(manifest, dev, False) if mk else _semseg(manifest, dev, 4)
    lr = mk(manifest, dev, True) if mk else _semseg(manifest, dev, 5)
    H, W = (64, 96) if kind == "psp" else (128, 256)
    clip = synth.make_clip(7, H, W, gop=3)
    frames = torch.from_numpy(clip["frames"]).to(dev)
    mvs = torch.from_numpy(clip["mv"]).to(dev)
    with torch.no_grad():
        full = hr(frames[0:1])
        out_k, feat_k = hr.forward_keyframe(frames[0:1])
        assert len(full) >= 3                                   # the reference's tuple, aux outputs included
        assert torch.equal(out_k, full[0]) and torch.equal(ops.as_nchw(feat_k), full[-1])
        x4 = ops.frame_ingest(frames[1:2], H // 2, W // 2, lr.storage_dtype)
        with_aux, without = lr.phase1_nhwc4(x4), lr.phase1_nhwc4(x4, aux=False)
        assert torch.equal(with_aux[-1], without[-1])
        ref_p = feat_k[0]
        a, pa = ev.alter_res_batch_fast(lr, [ref_p] * 2, frames[1:3], mvs[1:3], 0.5)
        prev = ops.configure(aux_outputs=True)
        try:
            b, pb = ev.alter_res_batch_fast(lr, [ref_p] * 2, frames[1:3], mvs[1:3], 0.5)
        finally:
            ops.configure(**prev)
        assert torch.equal(a, b) and torch.equal(pa, pb)


def test_gop_runner_single_gpu(dev, manifest):
    """GopRunner without a process group (world 1): keyframe -> exchange (no-op) -> batched non-keyframes."""
    from arseg_amd import evaluation as ev
    from arseg_amd import ops, synth
    from arseg_amd.gop import GopRunner

    hr, lr = _psp(manifest, dev, False), _psp(manifest, dev, True)
    clip = synth.make_clip(6, 48, 64, gop=12)
    frames = torch.from_numpy(clip["frames"]).to(dev)
    mvs = torch.from_numpy(clip["mv"]).to(dev)
    runner = GopRunner(lambda k: ops.to_nhwc(hr(k)[-1])[0],
                       lambda ref, img, mv: ev.alter_res_step_fast(lr, ref.unsqueeze(0), img, mv, 0.5)[0], n_gops=1, gop=12)
    assert runner.plan == [(0, d) for d in range(1, 12)]
    with torch.no_grad():
        out = runner.run_batched({0: frames[0:1]}, frames[1:12], mvs[1:12], lambda refs, imgs, mv: ev.alter_res_batch_fast(lr, refs, imgs, mv, 0.5)[0])
        single = runner.run({0: frames[0:1]}, {(0, d): frames[d:d + 1] for d in range(1, 12)}, {(0, d): mvs[d:d + 1] for d in range(1, 12)})
    assert out.shape == (11, 12, 48, 64)
    for d in range(1, 12):
        assert maxdiff(out[d - 1:d], single[(0, d)]) <= 2e-4


def test_eval_alter_res_keyframe_cache(dev, manifest):
    """EvalAlterRes(cache_keyframe=True): same mIoU as the reference-faithful evaluator, one HR forward per GOP."""
    from arseg_amd import evaluation as ev, synth

    hr, lr = _psp(manifest, dev, False), _psp(manifest, dev, True)
    clip = synth.make_clip(3, 48, 64, gop=4)
    key = torch.from_numpy(clip["frames"][0:1])
    g = np.random.Generator(np.random.PCG64(9))
    samples = []
    for d in (1, 2, 3):
        label = torch.from_numpy(g.integers(0, 12, (1, 48, 64)).astype(np.int64))
        flow = torch.from_numpy(clip["mv"][d:d + 1].astype(np.float64) / 4)
        samples.append((torch.from_numpy(clip["frames"][d:d + 1]), label, None, key.clone(), flow))
    with torch.no_grad():
        plain, cached = ev.EvalAlterRes(scale=0.5), ev.EvalAlterRes(scale=0.5, cache_keyframe=True)
        m0 = plain(hr, lr, samples, 12)
        m1 = cached(hr, lr, samples, 12)
    assert m0 == m1
    assert plain.hr_forwards == 3 and cached.hr_forwards == 1


def _nccl_worker(rank, world, port, q, mode):
    import os as _os

    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from arseg_amd import evaluation as ev
    from arseg_amd import ops, synth
    from arseg_amd.gop import GopRunner
    from arseg_amd.model import PSPNet, PSPNetWithFuse

    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ops.configure(conv_autotune=False)          # the same launch plans in every process: the comparison is bit for bit
    H, W = 64, 96
    hr = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
    lr = PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18", atten_k=7)
    synth.load_synth_weights(hr, 0)
    synth.load_synth_weights(lr, 1)
    hr, lr = hr.to(dev).eval(), lr.to(dev).eval()
    n_gops = 1 if mode in ("single", "loopback-broadcast") else world
    loop = {"loopback": "all_gather", "loopback-broadcast": "broadcast"}.get(mode, False)      # one rank, the collective still issued
    runner = GopRunner(lambda k: ops.to_nhwc(hr(k)[-1])[0], None, n_gops=n_gops, gop=12, loopback=loop)
    assert bool(runner.loopback) == bool(loop)
    if loop:
        runner.enable_timing()
    clips = {g: synth.make_clip(g, H, W, gop=12) for g in range(n_gops)}
    keyframes = {g: torch.from_numpy(clips[g]["frames"][0:1]).to(dev) for g in runner.my_gops}
    fb = torch.cat([torch.from_numpy(clips[g]["frames"][d:d + 1]) for g, d in runner.plan]).to(dev)
    mb = torch.cat([torch.from_numpy(clips[g]["mv"][d:d + 1]) for g, d in runner.plan]).to(dev)
    like = torch.empty((H, W, 64), dtype=torch.float32, device=dev)
    with torch.no_grad():
        out = runner.run_overlapped(keyframes, fb, mb, lambda f: ev.alter_res_phase1(lr, f, 0.5),
                                    lambda feat, refs, mvq: ev.alter_res_phase2(lr, feat, refs, mvq), like=like)
        hist = torch.tensor([float(out.shape[0])], device=dev)
        dist.all_reduce(hist)                                        # the confusion-matrix reduction (evaluation.py:134-135) on RCCL
    torch.cuda.synchronize()
    if loop:          # the side-stream collective ran (HIP events around it were recorded and can be read)
        st = runner.exchange_stats()
        assert st is not None and st["steps"] == 1 and st["exchange_ms"] > 0 and st["plan"] == ("broadcast" if loop == "broadcast" else "all_gather"), st
    q.put((rank, list(runner.plan), out.cpu().numpy().copy(), float(hist)))      # by value: a shared-memory handle dies with the worker
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["batched", "single", "loopback", "loopback-broadcast"])
def test_two_rank_rccl_matches_single_process(dev, mode):
    """The N-rank HIP path over RCCL (backend "nccl") equals the 1-rank path bit for bit: 2 ranks, the all-gather plan with the
    exchange overlapped with phase 1, and the single-GOP broadcast plan.  Needs two GPUs (skipped on a 1-GPU box).
    loopback / loopback-broadcast: ONE rank whose process group still issues the all-gather / broadcast on the side stream
    (GopRunner(loopback=...)) -- RCCL initialised on this box and the exchange code run through it; runs on a 1-GPU box."""
    import socket

    import torch.multiprocessing as mp

    world = 1 if mode.startswith("loopback") else 2
    if torch.cuda.device_count() < world:
        pytest.skip("needs >= 2 GPUs")
    from arseg_amd import evaluation as ev
    from arseg_amd import ops, synth
    from arseg_amd.model import PSPNet, PSPNetWithFuse

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference on this process' GPU: the same batches with the same launch plans
    prev = ops.configure(conv_autotune=False)
    try:
        H, W = 64, 96
        hr = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
        lr = PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18", atten_k=7)
        synth.load_synth_weights(hr, 0)
        synth.load_synth_weights(lr, 1)
        hr, lr = hr.to(dev).eval(), lr.to(dev).eval()
        n_gops = 1 if mode in ("single", "loopback-broadcast") else world
        clips = {g: synth.make_clip(g, H, W, gop=12) for g in range(n_gops)}
        seen = set()
        with torch.no_grad():
            refs = {g: ops.to_nhwc(hr(torch.from_numpy(clips[g]["frames"][0:1]).to(dev))[-1])[0] for g in range(n_gops)}
            for rank, plan, out, hist in results:
                assert hist == n_gops * 11
                fb = torch.cat([torch.from_numpy(clips[g]["frames"][d:d + 1]) for g, d in plan]).to(dev)
                mb = torch.cat([torch.from_numpy(clips[g]["mv"][d:d + 1]) for g, d in plan]).to(dev)
                want = ev.alter_res_phase2(lr, ev.alter_res_phase1(lr, fb, 0.5), [refs[g] for g, _ in plan], mb).cpu()
                assert np.array_equal(out, want.numpy()), rank
                seen.update(plan)
        assert len(seen) == n_gops * 11
    finally:
        ops.configure(**prev)


def _gloo_shared_gpu_worker(rank, world, port, q, mode):
    """One of `world` processes that share cuda:0 and exchange the keyframe features over gloo: the HIP path + the exchange buffers + the
    side-stream ordering of GopRunner.run_overlapped together, without a second GPU."""
    import os as _os

    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from arseg_amd import evaluation as ev
    from arseg_amd import ops, synth
    from arseg_amd.gop import GopRunner
    from arseg_amd.model import PSPNet, PSPNetWithFuse

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops.configure(conv_autotune=False)          # the same launch plans in every process: the comparison is bit for bit
    H, W = 64, 96
    hr = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
    lr = PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18", atten_k=7)
    synth.load_synth_weights(hr, 0)
    synth.load_synth_weights(lr, 1)
    hr, lr = hr.to(dev).eval(), lr.to(dev).eval()
    n_gops = 1 if mode == "single" else world
    runner = GopRunner(lambda k: ops.to_nhwc(hr(k)[-1])[0], None, n_gops=n_gops, gop=12, deal="neighbor" if mode == "neighbor" else "round_robin")
    clips = {g: synth.make_clip(g, H, W, gop=12) for g in range(n_gops)}
    keyframes = {g: torch.from_numpy(clips[g]["frames"][0:1]).to(dev) for g in runner.my_gops}
    fb = torch.cat([torch.from_numpy(clips[g]["frames"][d:d + 1]) for g, d in runner.plan]).to(dev)
    mb = torch.cat([torch.from_numpy(clips[g]["mv"][d:d + 1]) for g, d in runner.plan]).to(dev)
    like = torch.empty((H, W, 64), dtype=torch.float32, device=dev)
    outs = []
    with torch.no_grad():
        for _ in range(2):                      # twice on one lane: the second step reuses the lane's exchange buffer and side stream
            outs.append(runner.run_overlapped(keyframes, fb, mb, lambda f: ev.alter_res_phase1(lr, f, 0.5),
                                              lambda feat, refs, mvq: ev.alter_res_phase2(lr, feat, refs, mvq), like=like))
    torch.cuda.synchronize()
    q.put((rank, list(runner.plan), outs[0].cpu().numpy().copy(), outs[1].cpu().numpy().copy()))      # by value: a shared-memory handle dies with the worker
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["batched", "single", "neighbor"])
def test_two_ranks_on_one_gpu_gloo_match_single_process(dev, mode):
    """GopRunner.run_overlapped on the HIP path with the exchange over gloo, two processes sharing ONE GPU -- each rank's
    outputs equal, bit for bit, what a single process computes for the same frames (same batch, same launch plans).  Covers what the
    CPU gloo tests (toy functions) and the skipped RCCL test leave out on a 1-GPU box: exchange buffers, side stream, stream joins."""
    import socket

    import torch.multiprocessing as mp

    from arseg_amd import evaluation as ev
    from arseg_amd import ops, synth
    from arseg_amd.gop import frame_plan, neighbor_plan
    from arseg_amd.model import PSPNet, PSPNetWithFuse

    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_shared_gpu_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    prev = ops.configure(conv_autotune=False)
    try:
        H, W = 64, 96
        hr = PSPNet(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18")
        lr = PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=12, psp_size=512, deep_features_size=256, backend="resnet18", atten_k=7)
        synth.load_synth_weights(hr, 0)
        synth.load_synth_weights(lr, 1)
        hr, lr = hr.to(dev).eval(), lr.to(dev).eval()
        n_gops = 1 if mode == "single" else world
        clips = {g: synth.make_clip(g, H, W, gop=12) for g in range(n_gops)}
        plans = neighbor_plan(n_gops, 12, world) if mode == "neighbor" else frame_plan(n_gops, 12, world)      # (r6) the contiguous-run deal: one send / receive
        seen = set()
        with torch.no_grad():
            refs = {g: ops.to_nhwc(hr(torch.from_numpy(clips[g]["frames"][0:1]).to(dev))[-1])[0] for g in range(n_gops)}
            for rank, plan, out0, out1 in results:
                assert plan == plans[rank]
                fb = torch.cat([torch.from_numpy(clips[g]["frames"][d:d + 1]) for g, d in plan]).to(dev)
                mb = torch.cat([torch.from_numpy(clips[g]["mv"][d:d + 1]) for g, d in plan]).to(dev)
                want = ev.alter_res_phase2(lr, ev.alter_res_phase1(lr, fb, 0.5), [refs[g] for g, _ in plan], mb).cpu()
                assert np.array_equal(out0, want.numpy()) and np.array_equal(out1, want.numpy()), rank
                seen.update(plan)
        assert len(seen) == n_gops * 11
    finally:
        ops.configure(**prev)


@pytest.mark.parametrize("mode", ["rehearsal2", "rccl-loopback"])
def test_bench_multi_rank_path_end_to_end(dev, mode):
    """bench.py's N > 1 code path as the driver launches it, end to end, on the 1-GPU box: (rehearsal2) two ranks under torch.distributed.run that share
    the GPU and exchange over gloo (ARSEG_DIST_BACKEND=gloo) -- every rank-0-only block of the bench must be free of collectives, or this hangs: it
    did, in the in-step profile pass, until round 5 --; (rccl-loopback) one rank on RCCL with the collective issued (ARSEG_RCCL_LOOPBACK=1).  Both must end
    with exactly ONE line on stdout, the JSON, carrying the exchange diagnostics (RCCL's banner used to land behind it)."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if mode == "rehearsal2":
        env["ARSEG_DIST_BACKEND"] = "gloo"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2"]
    else:
        env.update(ARSEG_RCCL_LOOPBACK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    ex = d["exchange"]
    assert ex["plan"] == "all_gather" and ex["exchange_ms"] > 0 and len(ex["per_rank"]) == d["n_gpus"]
    assert d["value"] > 0 and d["unit"] == "frames/s" and d["executor"] == "eager launches"
    if mode == "rehearsal2":
        assert d["n_gpus"] == 2 and "rehearsal" in d and d["plans"]["local"]["value"] > 0 and "creff_compute_units" in d
    else:
        assert d["n_gpus"] == 1 and "rccl_loopback" in d and ex["backend"] == "nccl"
