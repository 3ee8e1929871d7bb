"""GPU parity tests of the 16-bit storage path (BASELINE configs[2]: BiSeNet-18 bf16, configs[4]: BiSeNet-18 0.3x fp16).

Op level: every 16-bit kernel against fp64 torch arithmetic on the SAME 16-bit-rounded operands -- what remains is fp32
accumulation order and the single output rounding (half an ulp: at most 2^-8 relative for bf16, 2^-11 for fp16).
Model level: the reference-generated fixtures G6 / G7 (fp32) with the measured error and the label agreement stated per dtype."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import maxdiff, t

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]
ULP = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}          # spacing relative to the bottom of a binade; one rounding to nearest errs <= ULP / 2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from arseg_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))


def close16(got, want, dtype, extra=0.0):
    """|got - want| <= ulp/2 * |want| + (fp32 accumulation slack) elementwise."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    tol = ULP[dtype] * 0.51 * want.abs() + extra + 1e-6
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), (float((got - want).abs().max()), int(bad.sum()))


CASES = [
    # N, H,  W,  Cin, Cout, k, stride, pad, dil, act,  bn,   bias,  res
    (1, 33, 47, 3, 64, 7, 2, 3, 1, "relu", True, False, False),      # stem (RGB padded to 8)
    (2, 20, 24, 64, 64, 3, 1, 1, 1, "relu", True, False, True),      # BasicBlock conv2 + residual
    (1, 17, 19, 64, 128, 3, 2, 1, 1, "relu", True, False, False),    # stride 2, 128-channel tile
    (1, 12, 16, 128, 128, 1, 2, 0, 1, "none", True, False, False),   # 1x1 stride-2 downsample
    (1, 9, 11, 512, 128, 1, 1, 0, 1, "relu", True, False, False),    # conv_avg-like 1x1, deep K
    (3, 1, 1, 128, 128, 1, 1, 0, 1, "sigmoid", True, False, False),  # attention vector (ARM / FFM)
    (1, 16, 24, 256, 256, 3, 1, 1, 1, "relu", True, False, False),   # feat_conv_out
    (1, 30, 40, 64, 72, 3, 1, 2, 2, "prelu", True, True, True),      # dilation, Cout not a multiple of the tile, bias
    (3, 37, 70, 128, 128, 3, 1, 1, 1, "relu", True, False, True),    # two 64-channel chunks, ragged 64-wide patch tiles, batch
    (1, 9, 150, 64, 192, 3, 1, 1, 1, "none", False, True, False),    # wide and flat, three 64-channel output tiles
    (2, 64, 96, 3, 64, 7, 2, 3, 1, "relu", True, False, False),      # stem, whole tiles
    (1, 18, 70, 3, 64, 7, 2, 3, 1, "prelu", False, True, False),     # stem, ragged in both directions
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[3]}to{c[4]}k{c[5]}s{c[6]}")
def test_conv2d16(dev, case, dtype):
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    N, H, W, Cin, Cout, k, stride, pad, dil, act, bn, bias, res = case
    x = rnd(1, N, Cin, H, W).to(dtype)
    w = rnd(2, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = rnd(3, Cout, scale=0.1) if bias else None
    bnp = None
    if bn:
        g = np.random.Generator(np.random.PCG64(4))
        bnp = (t(g.uniform(0.75, 1.25, Cout).astype(np.float32)), rnd(5, Cout, scale=0.1), rnd(6, Cout, scale=0.1),
               t(g.uniform(0.5, 1.5, Cout).astype(np.float32)))
    code = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "prelu": _lib.ACT_PRELU, "sigmoid": _lib.ACT_SIGMOID}[act]
    pc = PackedConv(w, b, bnp, stride, pad, dil, code, 0.2, dev)
    cpad = (Cin + 7) // 8 * 8
    xn = torch.zeros(N, H, W, cpad, dtype=dtype)
    xn[..., :Cin] = x.permute(0, 2, 3, 1)
    y = F.conv2d(x.double(), w.to(dtype).double(), None, stride=stride, padding=pad, dilation=dil)
    if bn:
        gam, bet, mu, var = (v.double() for v in bnp)
        sc = gam / torch.sqrt(var + 1e-5)
        sh = bet - mu * sc + (b.double() * sc if bias else 0)
        y = y * sc[None, :, None, None] + sh[None, :, None, None]
    elif bias:
        y = y + b.double()[None, :, None, None]
    r = None
    if res:
        r = rnd(7, *y.shape).to(dtype)
        y = y + r.double()
    y = {"none": lambda v: v, "relu": torch.relu, "prelu": lambda v: torch.where(v >= 0, v, 0.2 * v), "sigmoid": torch.sigmoid}[act](y)
    rd = None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dev)
    for cfg in (0, 1, 2, 3, 4):             # automatic choice, then every tile shape (64 / 128 channels x K step 32 / 64)
        got = ops.conv2d(xn.to(dev), pc, residual=rd, tile_cfg=cfg)
        assert got.dtype == dtype and got.shape == (N, y.shape[2], y.shape[3], Cout)
        close16(got.permute(0, 3, 1, 2), y, dtype, extra=2e-5 * float(y.abs().max()))
    if k == 7 and Cin == 3 and Cout == 64:                                # stem kernel
        got = ops.conv2d(xn.to(dev), pc, tile_cfg=9)
        close16(got.permute(0, 3, 1, 2), y, dtype, extra=2e-5 * float(y.abs().max()))
    if k == 3 and stride == 1 and pad == dil and Cin % 64 == 0:          # patch-resident plans
        for cfg in (5, 6, 7, 8):
            got = ops.conv2d(xn.to(dev), pc, residual=rd, tile_cfg=cfg)
            close16(got.permute(0, 3, 1, 2), y, dtype, extra=2e-5 * float(y.abs().max()))
        for cfg in (10, 11, 12, 13):          # (r6) the squarer pixel tiles; a map too narrow for one is refused, not mis-tiled
            try:
                got = ops.conv2d(xn.to(dev), pc, residual=rd, tile_cfg=cfg)
            except _lib.ArsegError as exc:
                assert "unsupported" in str(exc).lower() or "-2" in str(exc), exc
                continue
            close16(got.permute(0, 3, 1, 2), y, dtype, extra=2e-5 * float(y.abs().max()))
    if Cout % 8 == 0 and k * k * Cin >= 256:        # split-K: fp32 partial sums + the epilogue in the reduce kernel; deterministic
        for cfg, sk in ((1, 2), (3, 3), (4, 8)):
            got = ops.conv2d(xn.to(dev), pc, residual=rd, tile_cfg=cfg, split_k=sk)
            close16(got.permute(0, 3, 1, 2), y, dtype, extra=2e-5 * float(y.abs().max()))
            assert torch.equal(got, ops.conv2d(xn.to(dev), pc, residual=rd, tile_cfg=cfg, split_k=sk))


@pytest.mark.parametrize("dtype", DTYPES)
def test_small_layers16(dev, dtype):
    from arseg_amd import _lib, ops

    x = rnd(10, 2, 19, 26, 64).to(dtype)
    xd = x.to(dev)
    xc = x.double().permute(0, 3, 1, 2)
    # maxpool: exact (a selection)
    assert torch.equal(ops.maxpool3x3s2(xd).cpu(), F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).to(dtype))
    # global mean
    close16(ops.global_reduce(xd, _lib.REDUCE_MEAN)[:, 0, 0], xc.mean(dim=(2, 3)), dtype, extra=1e-6)
    # resize: nearest x2 exact, bilinear align_corners=True and False
    assert torch.equal(ops.resize_nhwc(xd, 38, 52, _lib.NEAREST, False).cpu(), F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0).permute(0, 2, 3, 1).to(dtype))
    for (Ho, Wo, al) in ((30, 45, True), (38, 52, False)):
        want = F.interpolate(xc, (Ho, Wo), mode="bilinear", align_corners=al)
        close16(ops.resize_nhwc(xd, Ho, Wo, _lib.BILINEAR, al).permute(0, 3, 1, 2), want, dtype, extra=1e-5)
    # scale_add
    sc, av, af = rnd(11, 2, 1, 1, 64).to(dtype), rnd(12, 2, 1, 1, 64).to(dtype), rnd(13, 2, 19, 26, 64).to(dtype)
    want = x.double() * sc.double() + af.double() + av.double()
    close16(ops.scale_add(xd, sc.to(dev), add_full=af.to(dev), add_vec=av.to(dev)), want, dtype, extra=1e-6)
    close16(ops.scale_add(xd, sc.to(dev)), x.double() * sc.double(), dtype)
    # head: fp32 logits
    wf, bf = rnd(14, 19, 64, scale=0.2), rnd(15, 19, scale=0.1)
    lg = ops.head(xd, wf.to(dev), bf.to(dev), log_softmax=False)
    assert lg.dtype == torch.float32 and maxdiff(lg, F.conv2d(xc, wf.double()[:, :, None, None], bf.double())) <= 1e-4
    for C2, ncls2, lsm in ((256, 19, True), (128, 12, False), (24, 5, True)):          # chunked matrix-core kernel (C % 64 == 0) / generic kernel
        x2 = rnd(17, 1, 9, 37, C2).to(dtype)
        wf2, bf2 = rnd(18, ncls2, C2, scale=0.2), rnd(19, ncls2, scale=0.1)
        want2 = F.conv2d(x2.double().permute(0, 3, 1, 2), wf2.double()[:, :, None, None], bf2.double())
        want2 = F.log_softmax(want2, dim=1) if lsm else want2
        assert maxdiff(ops.head(x2.to(dev), wf2.to(dev), bf2.to(dev), log_softmax=lsm), want2) <= 2e-4
    # frame ingest (+ downscale) and casts
    img = rnd(16, 2, 3, 36, 48)
    want = F.interpolate(img.double(), (18, 24), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    got = ops.frame_ingest(img.to(dev), 18, 24, dtype)
    assert got.shape == (2, 18, 24, 8) and float(got[..., 3:].abs().max()) == 0.0
    close16(got[..., :3], want, dtype, extra=1e-6)
    assert torch.equal(ops.cast(ops.cast(xd, torch.float32), dtype), xd) and torch.equal(ops.cast(xd, torch.float32).cpu(), x.float())
    assert torch.equal(ops.cast(rnd(17, 4, 40).to(dev), dtype).cpu(), rnd(17, 4, 40).to(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_warp_mvq16(dev, dtype):
    """MV resize + warp of a 16-bit keyframe feature (fp32 C8 out) against the oracle's warp of the same rounded feature."""
    from arseg_amd import _lib, ops
    from oracle import cpu_ref

    H, W, Hp, Wp, C = 64, 96, 8, 12, 64
    g = np.random.Generator(np.random.PCG64(9))
    mvq = torch.from_numpy((g.integers(-12, 13, (1, H, W, 2)) * 4).astype(np.int16))
    feat = rnd(10, 1, C, Hp, Wp).to(dtype)
    want = cpu_ref.warp_feature(feat.float(), cpu_ref.mv_resize(cpu_ref.mv_from_int16(mvq), Hp, Wp))
    out = torch.empty((1, C // 8, Hp, Wp, 8), dtype=torch.float32, device=dev)
    lib = _lib.load()
    import ctypes
    feat_d, mvq_d = feat.permute(0, 2, 3, 1).contiguous().to(dev), mvq.to(dev)          # (kept alive across the asynchronous launch)
    st = lib.arseg_warp_mvq16_fwd(ctypes.c_void_p(feat_d.data_ptr()), ops._DT16[dtype], ctypes.c_void_p(mvq_d.data_ptr()),
                                  ctypes.c_void_p(out.data_ptr()), 1, C, Hp, Wp, H, W, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    assert maxdiff(ops.from_c8(out, _lib.NCHW), want) <= 1e-5
    # one launch for the non-keyframes of a GOP (feature stride 0: every frame samples the same keyframe feature) == one launch per frame
    B = 3
    mvb = torch.from_numpy((g.integers(-12, 13, (B, H, W, 2)) * 4).astype(np.int16)).to(dev)
    outb = torch.empty((B, C // 8, Hp, Wp, 8), dtype=torch.float32, device=dev)
    st = lib.arseg_warp_mvq16_shared_fwd(ctypes.c_void_p(feat_d.data_ptr()), 0, ops._DT16[dtype], ctypes.c_void_p(mvb.data_ptr()),
                                         ctypes.c_void_p(outb.data_ptr()), B, C, Hp, Wp, H, W, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    for b in range(B):
        one = torch.empty((1, C // 8, Hp, Wp, 8), dtype=torch.float32, device=dev)
        assert lib.arseg_warp_mvq16_fwd(ctypes.c_void_p(feat_d.data_ptr()), ops._DT16[dtype], ctypes.c_void_p(mvb[b:b + 1].data_ptr()),
                                        ctypes.c_void_p(one.data_ptr()), 1, C, Hp, Wp, H, W, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        assert torch.equal(outb[b:b + 1], one)


def _bise16(manifest, dev, fuse, dtype):
    from arseg_amd import synth
    from arseg_amd.model import BiSeNetV1, BiSeNetV1WithFuse

    m = BiSeNetV1WithFuse(n_classes=12, backend="resnet18") if fuse else BiSeNetV1(n_classes=12, backend="resnet18")
    name, seed = ("BiSeNetV1WithFuse", 3) if fuse else ("BiSeNetV1", 2)
    spec = [(k, tuple(s)) for k, s in manifest[name]["keys"]]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, seed).items()})
    return m.to(dev).eval().set_storage(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bisenet16_golden(dev, golden, manifest, dtype):
    """BiSeNetV1 (HR branch) and BiSeNetV1WithFuse phase 1 / phase 2 with 16-bit tensors against the reference's fp32 fixtures
    (G6): reduced precision by construction -- stated: max-abs error relative to the tensor's magnitude and label agreement."""
    from arseg_amd import ops

    g = golden("g6_bisenet")
    hr = _bise16(manifest, dev, False, dtype)
    with torch.no_grad():
        out, o16, o32, fuse = hr(t(g["x"]).to(dev))
    assert out.dtype == torch.float32 and fuse.dtype == dtype and out.shape == g["out"].shape and fuse.shape == g["feat_fuse"].shape
    rel = {torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]              # measured <= 1.5e-3 / 1.2e-2 of the tensor's magnitude (x 2.5 margin)
    scale_f, scale_o = float(np.abs(g["feat_fuse"]).max()), float(np.abs(g["out"]).max())
    e_f, e_o = maxdiff(fuse.float(), g["feat_fuse"]), maxdiff(out, g["out"])
    agree = float((out.argmax(1).cpu().numpy() == g["out"].argmax(1)).mean())
    print(f"\\n[{dtype}] BiSeNetV1: feat_fuse err {e_f:.3e} (max {scale_f:.1f}), logits err {e_o:.3e} (max {scale_o:.1f}), labels equal {agree:.4f}")
    assert e_f <= rel * scale_f and e_o <= rel * scale_o
    assert agree >= {torch.float16: 0.998, torch.bfloat16: 0.99}[dtype]      # measured 0.9996-0.9998 / 0.9960-0.9976
    g2 = golden("g6_bisefuse")
    lr = _bise16(manifest, dev, True, dtype)
    with torch.no_grad():
        a16, a32, mid = lr.forward_phase1(t(g2["x"]).to(dev))
        ob, pb = lr.forward_phase2(mid, t(g2["ref_p"]).to(dev).to(dtype))
    e_m, e_ob = maxdiff(mid.float(), g2["mid"]), maxdiff(ob, g2["out"])
    print(f"[{dtype}] BiSeNetV1WithFuse: mid err {e_m:.3e} (max {float(np.abs(g2['mid']).max()):.1f}), logits err {e_ob:.3e} (max {float(np.abs(g2['out']).max()):.1f})")
    assert e_m <= rel * float(np.abs(g2["mid"]).max()) and e_ob <= rel * float(np.abs(g2["out"]).max())


@pytest.mark.parametrize("dtype", DTYPES)
def test_alter_res16_golden(dev, golden, manifest, dtype):
    """One EvalAlterRes step (G7, BiSeNet) on the 16-bit fast path: keyframe HR forward, LR backbone, MV warp of the 16-bit keyframe
    feature, fp32 CReFF, fused argmax tail; against the reference's fp32 logits / labels."""
    from arseg_amd import evaluation as ev
    from arseg_amd import ops

    g = golden("g7_alter_bise")
    hr, lr = _bise16(manifest, dev, False, dtype), _bise16(manifest, dev, True, dtype)
    img, ref, label, mvq = t(g["img"]), t(g["ref"]), t(g["label"]), t(g["mvq"])
    with torch.no_grad():
        ref_p = hr(ref.to(dev))[-1]
        assert ref_p.dtype == dtype
        out, _ = ev.alter_res_step_fast(lr, ops.to_nhwc(ref_p), img.to(dev), mvq.to(dev), 0.5)
        pred, hist = ev.alter_res_batch_pred(lr, [ops.to_nhwc(ref_p)[0]], img.to(dev), mvq.to(dev), 0.5, labels=label.to(dev))
    e = maxdiff(out, g["out"])
    agree = float((pred.cpu().long().numpy() == g["preds"]).mean())
    print(f"\\n[{dtype}] EvalAlterRes step: logits err {e:.3e} (max {float(np.abs(g['out']).max()):.1f}), labels equal {agree:.4f}")
    assert e <= {torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype] * float(np.abs(g["out"]).max())
    assert agree >= {torch.float16: 0.998, torch.bfloat16: 0.99}[dtype]      # measured 0.9996-0.9998 / 0.9960-0.9976
    assert int(hist.sum()) == int((label != 255).sum())


@pytest.mark.parametrize("dtype", DTYPES)
def test_bisenet16_odd_sizes_golden(dev, golden, manifest, dtype):
    """The odd-size path of BASELINE configs[4] (0.3x: 307x614 -> 40x78 maps; here the reference's 67x131 fixture G6-odd, whose
    16/32-stride maps need the re-interpolation branches of bisenet.py:298) with 16-bit storage, HR branch and LR phase 1."""
    go = golden("g6_biseodd")
    rel = {torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
    hr, lr = _bise16(manifest, dev, False, dtype), _bise16(manifest, dev, True, dtype)
    with torch.no_grad():
        oo, _, _, fo = hr(t(go["x"]).to(dev))
        a16, _, mid = lr.forward_phase1(t(go["x"]).to(dev))
    assert fo.dtype == dtype and mid.dtype == dtype and tuple(mid.shape) == go["mid"].shape
    e = {"hr_out": (maxdiff(oo[..., ::2, ::2], go["hr_out_s2"]), float(np.abs(go["hr_out_s2"]).max())),
         "hr_feat_fuse": (maxdiff(fo.float(), go["hr_feat_fuse"]), float(np.abs(go["hr_feat_fuse"]).max())),
         "mid": (maxdiff(mid.float(), go["mid"]), float(np.abs(go["mid"]).max())),
         "aux16": (maxdiff(a16[..., ::4, ::4], go["aux16_s4"]), float(np.abs(go["aux16_s4"]).max()))}
    print(f"\n[{dtype}] odd sizes: " + ", ".join(f"{k} err {v[0]:.3e} (max {v[1]:.1f})" for k, v in e.items()))
    for k, (err, mag) in e.items():
        assert err <= rel * mag, k


@pytest.mark.parametrize("dtype", DTYPES)
def test_phase2_warp16_noninteger_ratio(dev, manifest, dtype):
    """Warp + CReFF + head on 16-bit features at a non-integer LR/HR ratio (configs[4]: a 40x78 LR map under a 128x256 keyframe
    feature; here 10x18 under 21x37) against the oracle evaluated on the same rounded tensors: the stage itself computes in fp32,
    so the only difference to the oracle is the kernel's arithmetic (<= 2e-4 of the magnitude)."""
    from arseg_amd import ops
    from oracle import cpu_ref

    lr = _bise16(manifest, dev, True, dtype)
    sd = {k: v.detach().cpu().float() for k, v in lr.state_dict().items()}
    Hp, Wp, hp, wp, C = 21, 37, 10, 18, 256
    g = np.random.Generator(np.random.PCG64(17))
    mvq = torch.from_numpy((g.integers(-6, 7, (1, Hp * 8, Wp * 8, 2)) * 4).astype(np.int16))
    ref = rnd(31, 1, C, Hp, Wp).to(dtype)
    mid = rnd(32, 1, C, hp, wp).to(dtype)
    with torch.no_grad():
        lo, p_c8 = lr.phase2_warp(mid.permute(0, 2, 3, 1).contiguous().to(dev), [ref.permute(0, 2, 3, 1).contiguous().to(dev)[0]], mvq.to(dev),
                                  upsample=False)
        warped = cpu_ref.warp_feature(ref.float(), cpu_ref.mv_resize(cpu_ref.mv_from_int16(mvq), Hp, Wp))
        want_p = cpu_ref.my_attention(sd, "fuse_attention.", warped, mid.float(), 7, 7)
        want_lo = torch.nn.functional.conv2d(want_p, sd["conv_out.conv_out.weight"], sd["conv_out.conv_out.bias"])
    from arseg_amd import _lib
    e_p, e_l = maxdiff(ops.from_c8(p_c8, _lib.NCHW), want_p), maxdiff(lo, want_lo)
    print(f"\n[{dtype}] phase2_warp 10x18 -> 21x37: p err {e_p:.3e} (max {float(want_p.abs().max()):.1f}), head logits err {e_l:.3e}")
    assert e_p <= 2e-4 * float(want_p.abs().max()) and e_l <= 2e-4 * max(float(want_lo.abs().max()), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,W,h,w", [(64, 1200, 32, 600), (36, 48, 18, 24), (35, 47, 17, 23), (20, 32, 20, 32)])
def test_frame_ingest_16bit_equals_rounded_fp32(dev, dtype, H, W, h, w):
    """frame_ingest to NHWC8 fp16 / bf16 -- the row-staged kernel (W % 4 == 0, downscale; several 256-pixel segments per row in the first
    case), the per-pixel kernel (odd widths) and the copy case -- is the fp32 ingest (tests/test_gpu_ops.py pins that one to
    F.interpolate) rounded to the storage type: same taps, same blend (the fp32 value may differ in its last bit between two kernels --
    hipcc contracts the blend per kernel -- so the bound is one unit of the storage type, not equality)."""
    from arseg_amd import ops

    g = np.random.Generator(np.random.PCG64(91))
    img = torch.from_numpy(g.standard_normal((2, 3, H, W)).astype(np.float32)).to(dev)
    got = ops.frame_ingest(img, h, w, dtype)
    ref = ops.frame_to_nhwc4(img, h, w)
    assert got.shape == (2, h, w, 8) and got.dtype == dtype
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert bool(((got[..., :3].float() - ref[..., :3]).abs() <= ulp * ref[..., :3].abs() + 1e-7).all())
    assert float((got[..., :3].float() != ref[..., :3].to(dtype).float()).float().mean()) < 1e-3      # (and nearly always the same rounding)
    assert float(got[..., 3:].float().abs().max()) == 0.0


# Full-size bounds (the bench variants' max_abs_err moved 0.035 <-> 0.30 between runs with no test bounding it).  Measured on MI355X,
# round 5, over both plan sets the tuner produces (bounds = ~2x the worst measurement; the fp32 oracle's logits span about +-20):
#   max_abs = worst pixel of the x8-upsampled logits; rms = whole-map RMS error relative to the RMS of the oracle's logits; agree = label agreement
#   bf16 (0.5x): max_abs 0.33 (logits) / 0.26 (keyframe feature), rms 0.80 % / 0.79 %, agreement 0.9930 (bench clip: 0.9949-0.9956, max_abs 0.26-0.30)
#   fp16 (0.3x): max_abs 0.033 / 0.035, rms 0.088 % / 0.091 %, agreement 0.99945 -- the "0.035" of the round-4 records is this configuration's number
FULL16 = {
    # storage: (scale, max_abs bound, relative rms bound, label agreement bound)
    # (r6) tightened to ~1.5x what five rounds of boxes / launch plans measured (0.70 / 1.8e-2 / 0.985 were loose): bf16 0.29-0.33, 0.79-0.80 %,
    # 0.9930-0.9931; fp16 0.033-0.036, 0.088-0.091 %, 0.99941-0.99945
    "bf16": (0.5, 0.50, 1.2e-2, 0.990),
    "f16": (0.3, 0.055, 1.4e-3, 0.9988),
}


@pytest.mark.parametrize("storage", ["bf16", "f16"])
def test_full_size_16bit_step(dev, storage):
    """BASELINE configs[2] (bf16, LR 0.5x) and configs[4] (fp16, LR 0.3x = 307x614) at FULL size inside the suite (model/bisenet.py:546-575 at
    1024x2048): BiSeNet-18 keyframe HR forward + one non-keyframe through downscale -> LR backbone -> MV warp + CReFF (C = 256 at 128x256) + head,
    16-bit activations and weights, against the fp32 CPU oracle.  16-bit storage is reduced precision by construction, so the bounds are stated
    (FULL16) instead of 1e-3: worst pixel, relative RMS and label agreement of the logits, and the same for the keyframe feature; the fused
    argmax tail must agree with the argmax of the logits it never writes.  The odd LR size of configs[4] (307x614 -> 39x77 / 40x78 maps) is the
    path the reference re-interpolates on (bisenet.py:298,442)."""
    from arseg_amd import evaluation as ev
    from arseg_amd import ops, synth
    from arseg_amd.model import BiSeNetV1, BiSeNetV1WithFuse
    from oracle import cpu_ref
    import os

    scale, b_max, b_rms, b_agree = FULL16[storage]
    sdt = {"bf16": torch.bfloat16, "f16": torch.float16}[storage]
    H, W = 1024, 2048
    hr, lr = BiSeNetV1(n_classes=19, backend="resnet18"), BiSeNetV1WithFuse(n_classes=19, backend="resnet18")
    synth.load_synth_weights(hr, 0)
    synth.load_synth_weights(lr, 1)
    sd_hr = synth.resolve_aliases({k: v.clone() for k, v in hr.state_dict().items()})
    sd_lr = synth.resolve_aliases({k: v.clone() for k, v in lr.state_dict().items()})
    hr, lr = hr.to(dev).eval().set_storage(sdt), lr.to(dev).eval().set_storage(sdt)
    clip = synth.make_clip(0, H, W, gop=4, mean=synth.CITY_BISE_MEAN, std=synth.CITY_BISE_STD)
    key, img, mvq = (torch.from_numpy(clip[k][i:i + 1]) for k, i in (("frames", 0), ("frames", 3), ("mv", 3)))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        o_out, o_p, _, o_ref = cpu_ref.alter_res_step("bise", sd_hr, sd_lr, img, key, cpu_ref.mv_from_int16(mvq), scale)
        ref_p = hr(key.to(dev))[-1]
        out, _ = ev.alter_res_step_fast(lr, ops.to_nhwc(ref_p), img.to(dev), mvq.to(dev), scale)
        pred, _ = ev.alter_res_batch_pred(lr, [ops.to_nhwc(ref_p)[0]], img.to(dev), mvq.to(dev), scale)
    out, ref = out.float().cpu(), ref_p.float().cpu()
    stats = {"max_abs_logits": float((out - o_out).abs().max()), "rms_rel_logits": float((out - o_out).pow(2).mean().sqrt() / o_out.pow(2).mean().sqrt()),
             "max_abs_ref": float((ref - o_ref).abs().max()), "rms_rel_ref": float((ref - o_ref).pow(2).mean().sqrt() / o_ref.pow(2).mean().sqrt()),
             "agree": float((out.argmax(1) == o_out.argmax(1)).float().mean()), "agree_fused_tail": float((pred.cpu().long() == o_out.argmax(1)).float().mean()),
             "logit_span": float(o_out.abs().max()), "ref_span": float(o_ref.abs().max())}
    print("FULL16", storage, stats)
    assert stats["max_abs_logits"] <= b_max and stats["rms_rel_logits"] <= b_rms and stats["agree"] >= b_agree, stats
    assert stats["rms_rel_ref"] <= b_rms and stats["max_abs_ref"] <= b_max, stats
    assert stats["agree_fused_tail"] >= b_agree, stats
    # the fused tail computes the argmax of exactly the logits the un-fused path writes (same kernels up to the head): near-ties aside, identical
    same = float((pred.cpu().long() == out.argmax(1)).float().mean())
    assert same >= 0.9995, same
