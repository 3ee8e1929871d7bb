"""GPU parity tests, op level: every entry point of libarseg_hip.so against the oracle on seeded inputs
(the oracle -- oracle/cpu_ref.py, oracle/local_attn_ref.c, torch CPU functional ops -- is the checker only)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import maxdiff, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from arseg_amd import _lib

    _lib.load()          # the HIP library must be the thing under test: fail loudly if it is missing
    return torch.device("cuda:0")


def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))


# ---------------------------------------------------------------------------------------------- localAttention pair
@pytest.mark.parametrize("N,C,H,W,kH,kW", [(1, 8, 10, 12, 7, 7), (2, 5, 9, 33, 5, 5), (1, 3, 4, 5, 7, 7), (1, 16, 17, 70, 3, 3),
                                            (2, 6, 9, 8, 5, 7), (1, 64, 24, 40, 7, 7), (1, 1, 1, 1, 7, 7)])
def test_local_pair(dev, N, C, H, W, kH, kW):
    from arseg_amd import ops
    from oracle import c_ref, cpu_ref

    q, k = rnd(1, N, C, H, W), rnd(2, N, C, H, W)
    w = torch.softmax(rnd(3, N, H, W, kH * kW), dim=3)
    s = ops.local_similar(q.to(dev), k.to(dev), kH, kW).cpu()
    o = ops.local_weighting(k.to(dev), w.to(dev), kH, kW).cpu()
    assert maxdiff(s, cpu_ref.local_similar(q, k, kH, kW)) <= 1e-4
    assert maxdiff(o, cpu_ref.local_weighting(k, w, kH, kW)) <= 1e-5
    assert maxdiff(s, c_ref.local_similar(q.numpy(), k.numpy(), kH, kW)) <= 1e-4
    assert maxdiff(o, c_ref.local_weighting(k.numpy(), w.numpy(), kH, kW)) <= 1e-5
    # NHWC (channels_last) variants: same results, no layout change
    qc, kc = q.to(dev).contiguous(memory_format=torch.channels_last), k.to(dev).contiguous(memory_format=torch.channels_last)
    s2 = ops.local_similar(qc, kc, kH, kW).cpu()
    o2 = ops.local_weighting(kc, w.to(dev), kH, kW)
    assert maxdiff(s2, s) <= 1e-5 and maxdiff(o2.cpu(), o) <= 1e-6
    assert C == 1 or o2.is_contiguous(memory_format=torch.channels_last)


def test_local_pair_golden_and_shim(dev, golden):
    import arseg_amd.localAttention as la
    from arseg_amd.model import f_similar, f_weighting

    g = golden("g3_pair")
    kH, kW = int(g["kH"]), int(g["kW"])
    v, w, q = (t(g[k]).to(dev) for k in ("v", "w", "q"))
    assert maxdiff(la.weighting_forward(v, w, kH, kW), g["weighting_cpu"]) <= 1e-5      # reference's f_weighting_cpu
    assert maxdiff(la.similar_forward(q, v, kH, kW), g["similar_shim"]) <= 1e-5
    assert maxdiff(f_weighting(v, w, kH, kW), g["weighting_cpu"]) <= 1e-5
    assert maxdiff(f_similar(q, v, kH, kW), g["similar_shim"]) <= 1e-5
    with pytest.raises(NotImplementedError):
        la.similar_backward(q, v, kH, kW, True)


# ---------------------------------------------------------------------------------------------- warp / MVs
@pytest.mark.parametrize("seed", [0, 1])
def test_warp_golden(dev, golden, seed):
    from arseg_amd import evaluation as ev

    g = golden(f"g1_warp_s{seed}")
    feat = t(g["feat"]).to(dev)
    for k in ("int", "frac", "f32"):
        flow = t(g[f"flow_{k}"]).to(dev)
        assert maxdiff(ev.warpFeature(feat, flow), g[f"out_{k}"]) <= 1e-5                       # NCHW kernel
        cl = feat.contiguous(memory_format=torch.channels_last)
        out = ev.warpFeature(cl, flow)                                                          # NHWC kernel
        assert out.shape == feat.shape and maxdiff(out, g[f"out_{k}"]) <= 1e-5
    zero = torch.zeros(1, 12, 16, 2, dtype=torch.float64, device=dev)
    assert maxdiff(ev.warpFeature(feat, zero), g["out_zero"]) <= 1e-5


@pytest.mark.parametrize("C,H,W", [(64, 37, 53), (8, 5, 300), (256, 16, 32)])
def test_warp_large_motion(dev, C, H, W):
    from arseg_amd import _lib, ops
    from oracle import cpu_ref

    feat = rnd(5, 2, C, H, W)
    g = np.random.Generator(np.random.PCG64(6))
    flow = torch.from_numpy(g.uniform(-1.5 * W, 1.5 * W, (2, H, W, 2)))
    want = cpu_ref.warp_feature(feat, flow)
    got = ops.warp(feat.to(dev), flow.to(dev), _lib.NCHW).cpu()
    assert maxdiff(got, want) <= 2e-5
    nhwc = feat.permute(0, 2, 3, 1).contiguous().to(dev)
    got2 = ops.warp(nhwc, flow.to(dev), _lib.NHWC).permute(0, 3, 1, 2).cpu()
    assert maxdiff(got2, want) <= 2e-5
    got3 = ops.from_c8(ops.warp(nhwc, flow.float().to(dev), _lib.NHWC, _lib.C8), _lib.NCHW).cpu()
    assert maxdiff(got3, cpu_ref.warp_feature(feat, flow.float())) <= 2e-5


@pytest.mark.parametrize("seed", [0, 1])
def test_mv_resize_golden(dev, golden, seed):
    from arseg_amd import evaluation as ev
    from arseg_amd import ops

    g = golden(f"g2_mvresize_s{seed}")
    mvq = t(g["mvq"]).to(dev)
    for hp, wp in ((4, 6), (32, 48), (5, 7)):
        out = ops.mv_resize(mvq, hp, wp)
        assert out.dtype == torch.float64 and maxdiff(out, g[f"out_{hp}x{wp}"]) <= 1e-12
        out2 = ev.resize_flow(mvq.double() / 4, hp, wp)
        assert maxdiff(out2, g[f"out_{hp}x{wp}"]) <= 1e-12


@pytest.mark.parametrize("H,W,Hp,Wp,C", [(32, 48, 32, 48, 8), (64, 96, 8, 12, 16), (40, 56, 5, 7, 8)])
def test_warp_mvq_fused(dev, H, W, Hp, Wp, C):
    from arseg_amd import _lib, ops
    from oracle import cpu_ref

    g = np.random.Generator(np.random.PCG64(9))
    mvq = torch.from_numpy((g.integers(-12, 13, (1, H, W, 2)) * 4).astype(np.int16))
    feat = rnd(10, 1, C, Hp, Wp)
    want = cpu_ref.warp_feature(feat, cpu_ref.mv_resize(cpu_ref.mv_from_int16(mvq), Hp, Wp))
    nhwc = feat.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.from_c8(ops.warp_mvq(nhwc, mvq.to(dev), _lib.C8), _lib.NCHW).cpu()
    assert maxdiff(got, want) <= 1e-5
    got2 = ops.warp_mvq(nhwc, mvq.to(dev), _lib.NHWC).permute(0, 3, 1, 2).cpu()
    assert maxdiff(got2, want) <= 1e-5


# ---------------------------------------------------------------------------------------------- CReFF
@pytest.mark.parametrize("ci", range(5))
@pytest.mark.parametrize("seed", [0, 1])
def test_creff_golden(dev, golden, ci, seed):
    from arseg_amd.model import MyAttention

    g = golden(f"g3_attn_c{ci}_s{seed}")
    k = int(g["k"])
    C = g["hr"].shape[1]
    m = MyAttention(C, kW=k, kH=k)
    m.load_state_dict({kk[2:]: t(g[kk]) for kk in g.files if kk.startswith("w.")})
    m = m.to(dev).eval()
    out = m(t(g["hr"]).to(dev), t(g["lr"]).to(dev))
    assert out.shape == g["out"].shape
    assert maxdiff(out, g["out"]) <= 1e-4


@pytest.mark.parametrize("C,Hp,Wp,hp,wp,k,n_cls,logsm", [
    (64, 40, 70, 20, 35, 7, 12, True),      # PSPNet-like, ragged vs the 16x32 tile
    (64, 16, 32, 8, 16, 7, 12, True),       # exactly one tile
    (256, 19, 33, 10, 17, 7, 19, False),    # BiSeNet-like, small map -> short tiles
    (8, 5, 6, 5, 6, 7, 0, False),           # image smaller than the window, same-size lr
    (16, 33, 65, 11, 22, 5, 7, True),       # 5x5 window, generic class count
    (8, 18, 34, 9, 17, 3, 0, False),        # 3x3 window
    (32, 70, 40, 50, 80, 7, 32, False),     # lr larger than hr in one dim, 32 classes
])
@pytest.mark.parametrize("impl", ["valu", "mfma16", "mfma8"])
def test_creff_vs_oracle(dev, C, Hp, Wp, hp, wp, k, n_cls, logsm, impl, monkeypatch):
    """impl pins one of the two kernels of arseg_creff_fwd (fp32 VALU / split-fp16 matrix cores; the latter falls back to
    the former for shapes it does not cover, e.g. C % 16 != 0 or windows other than 7x7)."""
    from arseg_amd import _lib, ops, synth

    monkeypatch.setattr(ops.config, "creff_impl", impl[:4])
    if impl.startswith("mfma"):
        monkeypatch.setattr(ops.config, "creff_tile_rows", int(impl[4:]))          # tile height of the matrix-core kernel (16 / 8 rows)
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention
    from oracle import cpu_ref

    for N, gain in ((1, 0.35), (2, 1.0)):
        m = synth.load_synth_weights(MyAttention(C, kW=k, kH=k), 7, attn_gain=gain)
        sd = {kk: v.clone() for kk, v in m.state_dict().items()}
        hr, lr = rnd(20, N, C, Hp, Wp), rnd(21, N, C, hp, wp)
        want = cpu_ref.my_attention(sd, "", hr, lr, k, k)
        pa = PackedAttention(m, dev)
        head = None
        if n_cls:
            wf, bf = rnd(22, n_cls, C, scale=0.2), rnd(23, n_cls, scale=0.1)
            head = (wf.to(dev), bf.to(dev))
        p_c8, logits = ops.creff(ops.to_c8(hr.to(dev), _lib.NCHW), ops.to_nhwc(lr.to(dev)), pa, head, logsm, k, k)
        tol = 1e-4 if gain < 1 else 3e-4      # gain 1.0: near one-hot softmax amplifies fp32 rounding of the scores
        assert maxdiff(ops.from_c8(p_c8, _lib.NCHW), want) <= tol
        if n_cls:
            lg = F.conv2d(want, wf[:, :, None, None], bf)
            if logsm:
                lg = F.log_softmax(lg, dim=1)
            assert maxdiff(logits, lg) <= 2 * tol
        else:
            assert logits is None


def test_creff_padding_taps_take_softmax_mass(dev):
    """Border semantics: taps outside the image score 0 and still receive softmax mass.  With a zero
    query (all scores 0) the weights are uniform 1/49 over ALL taps, so a corner pixel keeps only 16/49 of a constant V."""
    from arseg_amd import _lib, ops
    from arseg_amd.model import MyAttention

    C = 8
    m = MyAttention(C, kW=7, kH=7)
    with torch.no_grad():
        for conv in (m.lr_query_conv, m.hr_key_conv, m.hr_value_conv):
            conv.weight.zero_()
            conv.bias.zero_()
        m.hr_value_conv.bias.fill_(1.0)          # V == 1 inside the image, 0 in the padding
    m = m.to(dev).eval()
    out = m(torch.zeros(1, C, 12, 40, device=dev), torch.zeros(1, C, 6, 20, device=dev)).cpu()
    assert abs(float(out[0, 0, 0, 0]) - 16 / 49) <= 1e-6
    assert abs(float(out[0, 0, 6, 20]) - 1.0) <= 1e-6
    assert abs(float(out[0, 0, 0, 20]) - 28 / 49) <= 1e-6


# ---------------------------------------------------------------------------------------------- conv engine
CONV_CASES = [
    # N, H,  W,  Cin, Cout, k, stride, pad, dil, act,  bn,    bias,  res
    (1, 33, 47, 3, 64, 7, 2, 3, 1, "relu", True, False, False),      # stem (RGB padded to 4)
    (1, 20, 24, 64, 64, 3, 1, 1, 1, "relu", True, False, True),      # BasicBlock conv2 + residual
    (2, 17, 19, 64, 128, 3, 2, 1, 1, "relu", True, False, False),    # stride 2
    (1, 12, 16, 128, 128, 1, 2, 0, 1, "none", True, False, False),   # 1x1 stride-2 downsample
    (1, 10, 12, 256, 256, 3, 1, 2, 2, "relu", True, False, True),    # dilation 2
    (1, 9, 11, 512, 512, 3, 1, 4, 4, "relu", True, False, False),    # dilation 4, deep K -> split-K
    (1, 8, 12, 2560, 1024, 1, 1, 0, 1, "relu", False, True, False),  # PSP bottleneck (Cin not a power of two)
    (1, 16, 24, 1024, 256, 3, 1, 1, 1, "prelu", True, True, False),  # up_1
    (1, 30, 40, 64, 12, 1, 1, 0, 1, "none", False, True, False),     # Cout not a multiple of 4
    (1, 1, 1, 256, 256, 1, 1, 0, 1, "sigmoid", True, False, False),  # attention vector (ARM / FFM)
    (3, 1, 1, 256, 12, 1, 1, 0, 1, "none", False, True, False),      # linear classifier
    (1, 40, 40, 64, 64, 3, 1, 1, 1, "prelu", True, True, False),     # up_3
    (2, 70, 131, 64, 96, 3, 1, 1, 1, "relu", True, False, True),     # ragged vs the 2x64 patch tiles, two images, Cout tail
]


def _conv_ref(x, w, b, bn, stride, pad, dil, act, slope, res):
    y = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
    if bn is not None:
        y = F.batch_norm(y, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    if res is not None:
        y = y + res
    if act == "relu":
        y = F.relu(y)
    elif act == "prelu":
        y = F.prelu(y, torch.tensor([slope], dtype=y.dtype))
    elif act == "sigmoid":
        y = torch.sigmoid(y)
    return y


@pytest.fixture(params=["f32", "f16x3"])
def conv_math(request):
    """Run a conv test under both MFMA back ends (fp32 MFMA; fp32 emulated by three fp16 MFMAs on split operands)."""
    from arseg_amd import ops

    prev = ops.set_conv_math(request.param)
    yield request.param
    ops.set_conv_math(prev)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"{c[3]}to{c[4]}k{c[5]}s{c[6]}d{c[8]}")
def test_conv2d(dev, case, conv_math):
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    N, H, W, Cin, Cout, k, stride, pad, dil, act, use_bn, use_bias, use_res = case
    g = np.random.Generator(np.random.PCG64(31))
    x = rnd(30, N, Cin, H, W)
    w = rnd(32, Cout, Cin, k, k, scale=float(np.sqrt(2.0 / (Cin * k * k))))
    b = rnd(33, Cout, scale=0.1) if use_bias else None
    bn = None
    if use_bn:
        bn = (t(g.uniform(0.5, 1.5, Cout).astype(np.float32)), rnd(34, Cout, scale=0.1), rnd(35, Cout, scale=0.1),
              t(g.uniform(0.5, 1.5, Cout).astype(np.float32)))
    slope = 0.2
    acts = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "prelu": _lib.ACT_PRELU, "sigmoid": _lib.ACT_SIGMOID}
    pc = PackedConv(w, b, bn, stride, pad, dil, acts[act], slope, dev)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = rnd(36, N, Cout, Ho, Wo) if use_res else None
    want = _conv_ref(x, w, b, bn, stride, pad, dil, act, slope, res)
    xn = x.permute(0, 2, 3, 1).contiguous()
    if Cin % 4:
        xn = F.pad(xn, (0, 4 - Cin % 4))
    xd = xn.to(dev)
    rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    tol = 2e-4
    for tile_cfg, split_k in ((0, 0), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1), (10, 1), (11, 1), (12, 1),
                              (3, 3), (7, 3), (1, 2), (5, 2), (11, 2), (9, 3), (13, 1), (14, 1), (15, 1), (16, 1), (17, 1), (18, 1), (19, 1), (19, 2)):
        if split_k > 1 and Cout % 4:
            continue
        if 13 <= tile_cfg <= 16 and not (conv_math == "f16x3" and k == 3 and stride == 1 and pad == dil and Cin % 32 == 0 and dil == 1):
            continue                  # the patch-resident kernel covers 3x3 stride-1 f16x3 convs only
        if tile_cfg >= 17 and conv_math != "f16x3":
            continue                  # the 8- / 16-wave tiles are built for the f16x3 back end only
        got = ops.conv2d(xd, pc, residual=rd, tile_cfg=tile_cfg, split_k=split_k).permute(0, 3, 1, 2).cpu()
        assert got.shape == want.shape
        assert maxdiff(got, want) <= tol, (tile_cfg, split_k)


@pytest.mark.parametrize("storage", ["f16", "bf16"])
def test_gemm_rows16(dev, storage):
    """arseg_gemm_rows16_fwd: the 1x1 convs of the 16-bit storage path (bisenet.py FFM / SpatialPath / ARM heads) on the LDS-DMA kernel -- every
    tile shape, with BN, bias, residual and ReLU, the output a channel slice -- against the same product of the rounded operands in fp64; and
    ops.conv2d picks whichever of this and the conv16 kernel is faster without changing the result beyond the output rounding."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    sdt = {"f16": torch.float16, "bf16": torch.bfloat16}[storage]
    N, H, W, Cin, Cout = 2, 9, 13, 128, 96
    g = np.random.Generator(np.random.PCG64(171))
    x = rnd(170, N, H, W, Cin).to(sdt)
    w = rnd(172, Cout, Cin, 1, 1, scale=0.1)
    bn = (t(g.uniform(0.5, 1.5, Cout).astype(np.float32)), rnd(174, Cout, scale=0.1), rnd(175, Cout, scale=0.1), t(g.uniform(0.5, 1.5, Cout).astype(np.float32)))
    res = rnd(176, N, H, W, Cout).to(sdt)
    pc = PackedConv(w, None, bn, 1, 0, 1, _lib.ACT_RELU, 0.0, dev)
    want = _conv_ref(x.double().permute(0, 3, 1, 2), w.to(sdt).double(), None, tuple(v.double() for v in bn), 1, 0, 1, "relu", 0.0, res.double().permute(0, 3, 1, 2))
    tol = float(want.abs().max()) * {"f16": 1.5e-3, "bf16": 1.2e-2}[storage]
    xd, rd = x.to(dev), res.to(dev)
    for cfg in range(12):
        wide = torch.zeros((N, H, W, Cout + 16), dtype=sdt, device=dev)
        ops.gemm_rows16(xd, pc, residual=rd, out=wide[..., 8:8 + Cout], cfg=cfg)
        assert maxdiff(wide[..., 8:8 + Cout].float().permute(0, 3, 1, 2), want) <= tol, cfg
        assert float(wide[..., :8].abs().max()) == 0 and float(wide[..., 8 + Cout:].abs().max()) == 0
    got = ops.conv2d(xd, pc, residual=rd)
    assert got.dtype == sdt and maxdiff(got.float().permute(0, 3, 1, 2), want) <= tol


def test_conv2d_channel_slices(dev, conv_math):
    """in_ld / out_ld: read a channel slice of a wider NHWC buffer and write into one (zero-copy concat)."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    x_big = rnd(40, 1, 9, 13, 96)                 # NHWC, use channels 32..95
    w = rnd(41, 32, 64, 3, 3, scale=0.05)
    pc = PackedConv(w, None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, dev)
    xd = x_big.to(dev)
    out_big = torch.full((1, 9, 13, 80), 7.0, device=dev)
    ops.conv2d(xd[..., 32:], pc, out=out_big[..., 16:48])
    want = F.conv2d(x_big[..., 32:].permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    assert maxdiff(out_big[..., 16:48], want) <= 1e-4
    assert float(out_big[..., :16].min()) == 7.0 and float(out_big[..., 48:].max()) == 7.0   # neighbours untouched


def test_conv2d_rejects_bad_arguments(dev):
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    pc = PackedConv(rnd(42, 8, 24, 3, 3), None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, dev)   # Cin=24: not a power of two with 3x3
    with pytest.raises(_lib.ArsegError):
        ops.conv2d(torch.zeros(1, 8, 8, 24, device=dev), pc)
    pc2 = PackedConv(rnd(43, 8, 16, 3, 3), None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, dev)
    with pytest.raises(_lib.ArsegError):
        ops.conv2d(torch.zeros(1, 8, 8, 16), pc2)                                            # CPU tensor: no fallback


# ---------------------------------------------------------------------------------------------- small layers
@pytest.mark.parametrize("H,W,C", [(33, 47, 64), (8, 8, 4), (1, 5, 128)])
def test_maxpool(dev, H, W, C):
    from arseg_amd import ops

    x = rnd(50, 2, C, H, W)
    got = ops.maxpool3x3s2(x.permute(0, 2, 3, 1).contiguous().to(dev)).permute(0, 3, 1, 2)
    assert maxdiff(got, F.max_pool2d(x, 3, 2, 1)) == 0.0


@pytest.mark.parametrize("N,H,W,C", [(1, 64, 128, 256), (2, 80, 70, 36), (1, 128, 256, 64), (11, 32, 64, 256)])
def test_global_reduce_two_stage(dev, N, H, W, C):
    """Global mean / max of large maps of few images: two deterministic stages through a workspace (row bands, then the bands in order); also a
    channel slice of a wider buffer, all-negative channels for the max, and run-to-run bit equality."""
    from arseg_amd import _lib, ops

    x = rnd(54, N, C, H, W)
    x[:, 0] = -x[:, 0].abs() - 1.0
    x[:, 3, 5, 7] = 1e30
    wide = torch.zeros(N, H, W, C + 4, device=dev)
    wide[..., 4:] = x.permute(0, 2, 3, 1).to(dev)
    mx = ops.global_reduce(wide[..., 4:], _lib.REDUCE_MAX).reshape(N, C)
    assert torch.equal(mx.cpu(), x.amax(dim=(2, 3)))
    x[:, 3, 5, 7] = 0.5
    wide[..., 4:] = x.permute(0, 2, 3, 1).to(dev)
    mean = ops.global_reduce(wide[..., 4:], _lib.REDUCE_MEAN).reshape(N, C)
    assert maxdiff(mean, x.double().mean(dim=(2, 3)).float()) <= 2e-6
    assert torch.equal(mean, ops.global_reduce(wide[..., 4:], _lib.REDUCE_MEAN).reshape(N, C))


@pytest.mark.parametrize("N,H,W,C,sizes", [(2, 32, 64, 512, (1, 2, 3, 6)), (1, 45, 90, 128, (1, 2, 3, 6)), (2, 9, 13, 68, (1, 2, 3, 6)), (3, 7, 5, 64, (2, 5)),
                                            (1, 4, 4, 32, (6,))])
def test_psp_pool_matrix(dev, N, H, W, C, sizes):
    """The folded pyramid's pooled matrix in one pass over the map (cell sums on the grid of all bin edges, then bins): every level's block equals
    F.adaptive_avg_pool2d (overlapping bins at non-divisible sizes, bins larger than the map), the sibling blocks are exact zeros."""
    from arseg_amd import ops

    x = rnd(53, N, C, H, W)
    got = ops.psp_pool_matrix(x.permute(0, 2, 3, 1).contiguous().to(dev), sizes).cpu()
    n, off = len(sizes), 0
    assert got.shape == (N, sum(s * s for s in sizes), 1, n * C)
    for i, s in enumerate(sizes):
        ref = F.adaptive_avg_pool2d(x, (s, s)).permute(0, 2, 3, 1).reshape(N, s * s, C)
        blk = got[:, off:off + s * s, 0, :].reshape(N, s * s, n, C)
        assert maxdiff(blk[:, :, i], ref) <= 1e-6
        others = blk.clone()
        others[:, :, i] = 0
        assert float(others.abs().max()) == 0.0
        off += s * s


@pytest.mark.parametrize("H,W,C,s", [(9, 12, 512, 1), (9, 12, 512, 2), (9, 12, 512, 3), (9, 12, 512, 6), (4, 5, 64, 6), (32, 64, 68, 3)])
def test_adaptive_avgpool_and_global_reduce(dev, H, W, C, s):
    from arseg_amd import _lib, ops

    x = rnd(51, 2, C, H, W)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.adaptive_avgpool(xd, s, s).permute(0, 3, 1, 2)
    assert maxdiff(got, F.adaptive_avg_pool2d(x, (s, s))) <= 1e-5
    wide = torch.zeros(2, H, W, C + 8, device=dev)
    wide[..., 8:] = xd
    assert maxdiff(ops.adaptive_avgpool(wide[..., 8:], s, s).permute(0, 3, 1, 2), F.adaptive_avg_pool2d(x, (s, s))) <= 1e-5
    assert maxdiff(ops.global_reduce(xd, _lib.REDUCE_MEAN).reshape(2, C), x.mean(dim=(2, 3))) <= 1e-5
    assert maxdiff(ops.global_reduce(xd, _lib.REDUCE_MAX).reshape(2, C), x.amax(dim=(2, 3))) == 0.0


@pytest.mark.parametrize("Hin,Win,Hout,Wout,mode,align", [
    (6, 8, 12, 16, "bilinear", False), (1, 1, 9, 12, "bilinear", False), (3, 3, 9, 12, "bilinear", False),
    (5, 7, 10, 14, "nearest", False), (10, 18, 9, 17, "bilinear", True), (9, 17, 10, 18, "bilinear", True),
    (16, 32, 128, 256, "bilinear", False), (8, 16, 128, 256, "bilinear", False), (48, 64, 24, 32, "bilinear", True),
    (7, 9, 7, 9, "bilinear", True), (5, 6, 13, 17, "bilinear", True),
    (3, 1, 24, 16, "bilinear", False), (5, 3, 17, 24, "bilinear", False), (2, 5, 32, 80, "bilinear", False)])      # run-based x8 / x16 kernel: one column, ragged rows
def test_resize(dev, Hin, Win, Hout, Wout, mode, align):
    from arseg_amd import _lib, ops

    x = rnd(52, 2, 12, Hin, Win)
    kw = dict(mode=mode) if mode == "nearest" else dict(mode=mode, align_corners=align)
    want = F.interpolate(x, (Hout, Wout), **kw)
    m = _lib.NEAREST if mode == "nearest" else _lib.BILINEAR
    got = ops.resize_nchw(x.to(dev), Hout, Wout, m, align)
    assert maxdiff(got, want) <= 1e-5
    got2 = ops.resize_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), Hout, Wout, m, align).permute(0, 3, 1, 2)
    assert maxdiff(got2, want) <= 1e-5


def test_scale_add_head_frame_layouts(dev):
    from arseg_amd import _lib, ops

    x = rnd(53, 2, 7, 9, 16)
    sc, av, af = rnd(54, 2, 1, 1, 16), rnd(55, 2, 1, 1, 16), rnd(56, 2, 7, 9, 16)
    assert maxdiff(ops.scale_add(x.to(dev), sc.to(dev), add_vec=av.to(dev)), x * sc + av) <= 1e-6
    assert maxdiff(ops.scale_add(x.to(dev), sc.to(dev), add_full=af.to(dev)), x * sc + af) <= 1e-6
    # head
    for n_cls, logsm in ((12, True), (19, False), (5, True), (32, False)):
        p = rnd(57, 2, 11, 13, 64)
        wf, bf = rnd(58, n_cls, 64, scale=0.2), rnd(59, n_cls, scale=0.1)
        want = F.conv2d(p.permute(0, 3, 1, 2), wf[:, :, None, None], bf)
        if logsm:
            want = F.log_softmax(want, dim=1)
        assert maxdiff(ops.head(p.to(dev), wf.to(dev), bf.to(dev), logsm), want) <= 1e-4
    for C, n_cls, logsm in ((128, 19, True), (8, 3, True), (20, 12, False)):          # wider / narrower features; C % 8 != 0 takes the generic kernel
        p = rnd(157, 1, 9, 37, C)
        wf, bf = rnd(158, n_cls, C, scale=0.2), rnd(159, n_cls, scale=0.1)
        want = F.conv2d(p.permute(0, 3, 1, 2), wf[:, :, None, None], bf)
        want = F.log_softmax(want, dim=1) if logsm else want
        assert maxdiff(ops.head(p.to(dev), wf.to(dev), bf.to(dev), logsm), want) <= 1e-4
    # frame ingest with and without downscale
    img = rnd(60, 2, 3, 20, 30)
    same = ops.frame_to_nhwc4(img.to(dev), 20, 30).cpu()
    assert maxdiff(same[..., :3], img.permute(0, 2, 3, 1)) == 0.0 and float(same[..., 3].abs().max()) == 0.0
    small = ops.frame_to_nhwc4(img.to(dev), 10, 15).cpu()
    assert maxdiff(small[..., :3].permute(0, 3, 1, 2), F.interpolate(img, (10, 15), mode="bilinear", align_corners=True)) <= 1e-5
    # layout round trips
    y = rnd(61, 2, 24, 5, 7)
    nhwc = ops.to_nhwc(y.to(dev))
    assert maxdiff(nhwc, y.permute(0, 2, 3, 1)) == 0.0
    assert maxdiff(ops.to_nchw_contiguous(nhwc), y) == 0.0
    assert maxdiff(ops.from_c8(ops.to_c8(y.to(dev), _lib.NCHW), _lib.NCHW), y) == 0.0
    assert maxdiff(ops.from_c8(ops.to_c8(nhwc, _lib.NHWC), _lib.NHWC), y.permute(0, 2, 3, 1)) == 0.0
    assert maxdiff(ops.from_c8(ops.to_c8(y.to(dev), _lib.NCHW), _lib.NHWC), y.permute(0, 2, 3, 1)) == 0.0


@pytest.mark.parametrize("h,w,H,W", [(12, 16, 12, 16), (6, 8, 12, 16), (24, 32, 13, 17)])
def test_argmax_confusion(dev, h, w, H, W):
    from arseg_amd import ops
    from oracle import cpu_ref

    n_cls = 12
    logits = rnd(62, 2, n_cls, h, w)
    g = np.random.Generator(np.random.PCG64(63))
    label = torch.from_numpy(g.integers(0, n_cls, (2, H, W)).astype(np.int64))
    label[0, :2, :3] = 255
    preds, hist = cpu_ref.eval_tail(logits, label, n_cls)
    got_p, got_h = ops.argmax_confusion(logits.to(dev), label.to(dev), H, W)
    if (h, w) == (H, W):
        # integer / index work at the logits' own size: bit-exact against the reference's argmax(softmax(.)) and its histogram
        assert torch.equal(got_p.cpu().long(), torch.argmax(torch.softmax(logits, dim=1), dim=1))
        assert torch.equal(got_p.cpu().long(), preds)
        assert torch.equal(got_h.cpu(), hist.long())
    else:
        # resized logits: the bilinear arithmetic differs in rounding, labels may flip only where the top two classes tie to ~1e-6
        lg = F.interpolate(logits, (H, W), mode="bilinear", align_corners=True)
        top2 = lg.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-5
        assert torch.equal(got_p.cpu().long()[clear], preds[clear])
        assert float((got_h.cpu().float() - hist).abs().sum()) <= 2 * int((~clear).sum()) + 0
    got_p2, got_h2 = ops.argmax_confusion(logits.to(dev), label.to(dev), H, W, hist=got_h.clone())
    assert torch.equal(got_h2, 2 * got_h)                              # accumulates, deterministic


def test_argmax_ties_and_nan(dev):
    """torch.argmax semantics: the first maximum wins a tie; a NaN counts as the maximum (first NaN wins)."""
    from arseg_amd import ops

    logits = torch.zeros(1, 5, 2, 4)
    logits[0, :, 0, 0] = torch.tensor([1.0, 3.0, 3.0, 2.0, 3.0])                 # three-way tie -> 1
    logits[0, :, 0, 1] = torch.tensor([0.0, float("nan"), 9.0, float("nan"), 1.0])   # first NaN -> 1
    logits[0, :, 0, 2] = torch.tensor([-1.0, -2.0, -0.5, -0.5, -3.0])             # tie of negatives -> 2
    logits[0, :, 0, 3] = torch.tensor([float("-inf")] * 5)                        # all -inf -> 0
    logits[0, :, 1, 0] = torch.tensor([float("inf"), 1.0, float("inf"), 0.0, 0.0])   # tie of +inf -> 0
    want = torch.argmax(logits, dim=1)
    label = torch.zeros(1, 2, 4, dtype=torch.int64)
    got, hist = ops.argmax_confusion(logits.to(dev), label.to(dev), 2, 4)
    assert torch.equal(got.cpu().long(), want)
    assert int(hist.sum()) == 8 and torch.equal(hist.cpu()[0], torch.bincount(want.flatten(), minlength=5))


@pytest.mark.parametrize("h,w,up", [(16, 24, 8), (5, 7, 8), (9, 11, 2), (33, 65, 4), (1, 1, 8), (40, 3, 8), (7, 5, 3)])
def test_argmax_fused_upsample(dev, h, w, up):
    """f3: head logits -> x`up` bilinear (align_corners=False, BiSeNetOutput.up, model/bisenet.py:215-216) -> argmax -> confusion
    without materialising the full-resolution logits, against torch on the CPU."""
    from arseg_amd import ops

    n_cls = 19
    logits = rnd(64, 2, n_cls, h, w)
    H, W = up * h, up * w
    g = np.random.Generator(np.random.PCG64(65))
    label = torch.from_numpy(g.integers(0, n_cls, (2, H, W)).astype(np.int64))
    label[1, -3:, :] = 255
    full = F.interpolate(logits, scale_factor=float(up), mode="bilinear", align_corners=False)
    want = torch.argmax(torch.softmax(full, dim=1), dim=1)
    got, hist = ops.argmax_confusion(logits.to(dev), label.to(dev), H, W, align_corners=False)
    top2 = full.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(got.cpu().long()[clear], want[clear])
    keep = label != 255
    want_h = torch.bincount(label[keep] * n_cls + want[keep], minlength=n_cls * n_cls).view(n_cls, n_cls)
    assert float((hist.cpu() - want_h).abs().sum()) <= 2 * int((~clear).sum())


@pytest.mark.parametrize("N,H,W,Cin,Cout,dil,act,use_res", [(2, 19, 26, 64, 64, 1, "relu", True), (1, 32, 64, 128, 96, 2, "prelu", False),
                                                           (1, 17, 33, 256, 64, 4, "none", True), (3, 8, 8, 64, 128, 1, "relu", False)])
def test_conv2d_winograd(dev, N, H, W, Cin, Cout, dil, act, use_res, conv_math):
    """Winograd F(4x4,3x3) path (incl. the polyphase handling of dilation and ragged tiles) against F.conv2d."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    g = np.random.Generator(np.random.PCG64(77))
    x = rnd(70, N, Cin, H, W)
    w = rnd(71, Cout, Cin, 3, 3, scale=float(np.sqrt(2.0 / (Cin * 9))))
    bn = (t(g.uniform(0.5, 1.5, Cout).astype(np.float32)), rnd(72, Cout, scale=0.1), rnd(73, Cout, scale=0.1),
          t(g.uniform(0.5, 1.5, Cout).astype(np.float32)))
    acts = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "prelu": _lib.ACT_PRELU}
    pc = PackedConv(w, None, bn, 1, dil, dil, acts[act], 0.2, dev)
    assert pc.wino_u is not None
    res = rnd(74, N, Cout, H, W) if use_res else None
    want = _conv_ref(x, w, None, bn, 1, dil, dil, act, 0.2, res)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty((N, H, W, Cout), device=dev)
    ops._conv_wino(xd, pc, rd, out, N, H, W)
    assert maxdiff(out.permute(0, 3, 1, 2), want) <= 2e-4
    direct = ops.conv2d(xd, pc, residual=rd, tile_cfg=7, split_k=1)
    assert maxdiff(direct, out) <= 2e-4


@pytest.mark.parametrize("N,H,W,Cin,Cout,dil,up2", [(2, 19, 26, 64, 64, 1, False), (1, 32, 64, 128, 96, 2, False), (1, 17, 33, 256, 36, 4, False),
                                                    (1, 40, 72, 512, 512, 1, False), (2, 18, 28, 64, 128, 1, True)])
def test_conv2d_winograd_gemm_x3(dev, N, H, W, Cin, Cout, dil, up2):
    """The Winograd route with its 36 GEMMs on the LDS-DMA kernel (split-row transform + arseg_gemm_x3_fwd, every tile_cfg) and the same
    route on the implicit-GEMM kernel against F.conv2d (both hold the Winograd tolerance; the x3 tile shapes agree with each other)."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    g = np.random.Generator(np.random.PCG64(177))
    h, w_ = (H // 2, W // 2) if up2 else (H, W)
    x = rnd(170, N, Cin, h, w_)
    w = rnd(171, Cout, Cin, 3, 3, scale=float(np.sqrt(2.0 / (Cin * 9))))
    bn = (t(g.uniform(0.5, 1.5, Cout).astype(np.float32)), rnd(172, Cout, scale=0.1), rnd(173, Cout, scale=0.1),
          t(g.uniform(0.5, 1.5, Cout).astype(np.float32)))
    pc = PackedConv(w, None, bn, 1, dil, dil, _lib.ACT_PRELU, 0.2, dev)
    xin = F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False) if up2 else x
    want = _conv_ref(xin, w, None, bn, 1, dil, dil, "prelu", 0.2, None)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    T = _lib.load().arseg_wino43_tiles(N, H, W, dil)
    key = ("wino_gemm", dev.index, T, pc.cin_pad, Cout, _lib.MATH_F16X3)
    prev_math = ops.set_conv_math("f16x3")
    saved = ops._conv_plans.get(key)
    try:
        outs = {}
        for plan in (7, 100, 101, 102, 103, 104, 105, 106):
            ops._conv_plans[key] = plan
            out = torch.full((N, H, W, Cout), float("nan"), device=dev)
            ops._conv_wino(xd, pc, None, out, N, H, W, up2=up2)
            outs[plan] = out.permute(0, 3, 1, 2).cpu()
            assert maxdiff(outs[plan], want) <= 2e-4, plan
        for plan in range(100, 107):
            assert maxdiff(outs[plan], outs[100]) <= 1e-6                # every tile shape adds the K steps in the same order
    finally:
        ops.set_conv_math(prev_math)
        if saved is None:
            ops._conv_plans.pop(key, None)
        else:
            ops._conv_plans[key] = saved


@pytest.mark.parametrize("N,h,w,Cin,Cmid,Cout", [(2, 9, 13, 64, 128, 64), (1, 16, 32, 512, 1024, 256), (1, 5, 7, 96, 32, 12)])
def test_conv_chain_split_rows(dev, N, h, w, Cin, Cmid, Cout):
    """The PSP bottleneck -> up_1 chain on split rows: a 1x1 conv (+ residual, ReLU) on the LDS-DMA GEMM writing ops.SplitRows, consumed by the
    tap-decomposed conv3x3-after-upsample (its low-resolution GEMM stages the split rows directly) -- against F.conv2d / F.interpolate; plus
    the same 1x1 conv with fp32 output (split pre-pass route), SplitRows.float(), and a consumer that has to fall back to the fp32 form."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    prev = ops.set_conv_math("f16x3")
    try:
        assert ops.gemm_x3_enabled()
        x = rnd(400, N, Cin, h, w)
        w1 = rnd(401, Cmid, Cin, 1, 1, scale=float(np.sqrt(2.0 / Cin)))
        b1 = rnd(402, Cmid, scale=0.1)
        res = rnd(403, N, Cmid, h, w)
        w3 = rnd(404, Cout, Cmid, 3, 3, scale=float(np.sqrt(2.0 / (9 * Cmid))))
        b3 = rnd(405, Cout, scale=0.1)
        pc1 = PackedConv(w1, b1, None, 1, 0, 1, _lib.ACT_RELU, 0.0, dev)
        pc3 = PackedConv(w3, b3, None, 1, 1, 1, _lib.ACT_PRELU, 0.25, dev)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        rd = res.permute(0, 2, 3, 1).contiguous().to(dev)
        mid_want = F.relu(F.conv2d(x.double(), w1.double(), b1.double()) + res.double())
        mid = ops.conv2d(xd, pc1, residual=rd, out_split=True)
        assert isinstance(mid, ops.SplitRows)
        assert maxdiff(mid.float().permute(0, 3, 1, 2), mid_want.float()) <= 2e-5
        plain = ops._conv1x1_x3(xd, pc1, rd)                       # fp32 output
        assert maxdiff(plain.permute(0, 3, 1, 2), mid_want.float()) <= 2e-5
        assert maxdiff(plain, mid.float()) <= 1e-6 * float(mid_want.abs().max())      # the split rows carry 22 bits of the same values
        up = F.interpolate(mid_want, scale_factor=2.0, mode="bilinear", align_corners=False)
        want = F.prelu(F.conv2d(up, w3.double(), b3.double(), padding=1), torch.tensor([0.25], dtype=torch.float64)).float()
        got = ops.conv2d(mid, pc3, up2=True)
        assert maxdiff(got.permute(0, 3, 1, 2), want) <= 5e-5
        # up_1 -> up_2: the gather writes split rows as well and a second tap-decomposed upsample conv consumes them
        if Cout % 32 == 0:
            w4 = rnd(406, 32, Cout, 3, 3, scale=float(np.sqrt(2.0 / (9 * Cout))))
            pc4 = PackedConv(w4, None, None, 1, 1, 1, _lib.ACT_PRELU, 0.1, dev)
            mid2 = ops.conv2d(mid, pc3, up2=True, out_split=True)
            assert isinstance(mid2, ops.SplitRows) and maxdiff(mid2.float(), got) <= 1e-6 * float(got.abs().max())
            up2_ = F.interpolate(want.double(), scale_factor=2.0, mode="bilinear", align_corners=False)
            want2 = F.prelu(F.conv2d(up2_, w4.double(), None, padding=1), torch.tensor([0.1], dtype=torch.float64)).float()
            got2 = ops.conv2d(mid2, pc4, up2=True)
            assert maxdiff(got2.permute(0, 3, 1, 2), want2) <= 1e-4
        got_fp32_in = ops.conv2d(plain, pc3, up2=True)             # whatever plan the tuner picks for the fp32 input
        assert maxdiff(got_fp32_in, got) <= 1e-4
        # a consumer without a split-row route (3x3 at the same resolution) reads the fp32 form
        same = ops.conv2d(mid, pc3)
        want_same = F.prelu(F.conv2d(mid_want, w3.double(), b3.double(), padding=1), torch.tensor([0.25], dtype=torch.float64)).float()
        assert maxdiff(same.permute(0, 3, 1, 2), want_same) <= 2e-4
        # with the route switched off the same call returns an ordinary tensor
        old = ops.configure(conv_gemm_x3=False)
        try:
            t_plain = ops.conv2d(xd, pc1, residual=rd, out_split=True)
            assert isinstance(t_plain, torch.Tensor) and maxdiff(t_plain.permute(0, 3, 1, 2), mid_want.float()) <= 2e-5
        finally:
            ops.configure(**old)
    finally:
        ops.set_conv_math(prev)


@pytest.mark.parametrize("B,M,K,N", [(1, 1, 32, 4), (3, 300, 96, 36), (2, 257, 64, 260), (1, 1000, 256, 512), (36, 130, 128, 64)])
def test_gemm_x3(dev, B, M, K, N):
    """arseg_split_rows_fwd + arseg_gemm_x3_fwd (every tile_cfg; ragged M and N tails, single-step K, scale / bias / activation epilogue)
    against an fp64 matmul: fp32-grade results from three fp16 products."""
    import ctypes

    from arseg_amd import _lib

    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda a: ctypes.c_void_p(a.data_ptr() if a is not None else None)      # noqa: E731
    g = np.random.Generator(np.random.PCG64(300 + M))
    x = torch.from_numpy((g.standard_normal((B, M, K)) * np.exp(g.standard_normal((B, M, 1)))).astype(np.float32)).to(dev)
    w = rnd(301, B, N, K, scale=0.1).to(dev)
    scale, bias = torch.from_numpy(g.uniform(0.5, 1.5, N).astype(np.float32)).to(dev), rnd(302, N).to(dev)
    xs, ws = torch.empty_like(x), torch.empty_like(w)
    _lib.check(lib.arseg_split_rows_fwd(P(x), K, P(xs), B * M, K, 1.0, None, 0.0, st), "split")
    _lib.check(lib.arseg_split_rows_fwd(P(w), K, P(ws), B * N, K, 1.0, None, 0.0, st), "split")
    # the split rows hold hi + lo = x to 22 bits
    raw = xs.view(torch.float16).view(B, M, K // 32, 2, 32).float()
    assert float((raw[:, :, :, 0] + raw[:, :, :, 1] - x.view(B, M, K // 32, 32)).abs().max()) <= 2.0 ** -21 * float(x.abs().max())
    ref = torch.bmm(x.double(), w.double().transpose(1, 2))
    for act, slope in ((_lib.ACT_NONE, 0.0), (_lib.ACT_PRELU, 0.25)):
        want = ref if act == _lib.ACT_NONE else F.prelu(ref * scale.double() + bias.double(), torch.tensor([slope], dtype=torch.float64, device=dev))
        for cfg in range(12):          # 0-6: the GEMM tiles of rounds 3-4, 7-11: the narrow tiles added for the implicit 3x3 convs (r5)
            out = torch.full((B, M, N), float("nan"), device=dev)
            sb = (None, None) if act == _lib.ACT_NONE else (scale, bias)
            _lib.check(lib.arseg_gemm_x3_fwd(P(xs), P(ws), P(out), M, N, K, N, B, M * K * 4, N * K * 4, M * N, P(sb[0]), P(sb[1]), None, 0, act, slope, 0, cfg, None, 0.0, st), "gemm_x3")
            assert float((out.double() - want).abs().max()) <= 3e-6 * float(want.abs().max()), (act, cfg)
    assert lib.arseg_gemm_x3_fwd(P(xs), P(ws), P(out), M, N, K + 1, N, B, 0, 0, 0, None, None, None, 0, 0, 0.0, 0, 0, None, 0.0, st) == _lib.ARSEG_EINVAL
    assert lib.arseg_gemm_x3_fwd(P(xs), P(ws), P(out), M, N, K, N, B, 0, 0, 0, None, None, None, 0, 0, 0.0, 0, 12, None, 0.0, st) == _lib.ARSEG_EINVAL


@pytest.mark.parametrize("N,h,w,Cin,Cout", [(2, 9, 13, 64, 64), (1, 16, 32, 256, 64), (1, 1, 1, 64, 32), (3, 33, 70, 64, 128), (2, 7, 40, 64, 64)])
def test_conv2d_fused_upsample(dev, N, h, w, Cin, Cout, conv_math):
    """PSPUpsample: F.upsample(x2, bilinear, align_corners=False) -> conv3x3 (+BN+PReLU), upsample fused into the Winograd
    input transform (borders included) vs materialised + direct."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    g = np.random.Generator(np.random.PCG64(81))
    x = rnd(80, N, Cin, h, w)
    wt = rnd(82, Cout, Cin, 3, 3, scale=float(np.sqrt(2.0 / (Cin * 9))))
    b = rnd(83, Cout, scale=0.1)
    bn = (t(g.uniform(0.5, 1.5, Cout).astype(np.float32)), rnd(84, Cout, scale=0.1), rnd(85, Cout, scale=0.1),
          t(g.uniform(0.5, 1.5, Cout).astype(np.float32)))
    pc = PackedConv(wt, b, bn, 1, 1, 1, _lib.ACT_PRELU, 0.3, dev)
    up = F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
    want = _conv_ref(up, wt, b, bn, 1, 1, 1, "prelu", 0.3, None)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty((N, 2 * h, 2 * w, Cout), device=dev)
    ops._conv_wino(xd, pc, None, out, N, 2 * h, 2 * w, up2=True)
    assert maxdiff(out.permute(0, 3, 1, 2), want) <= 2e-4
    got = ops.conv2d(xd, pc, up2=True)                      # autotuned choice
    assert maxdiff(got.permute(0, 3, 1, 2), want) <= 2e-4
    # tap decomposition (csrc/upconv.hip): nine 1x1 taps at low resolution + the gather over the upsampled positions; no transform, so it is
    # held to the direct form's tolerance
    outt = torch.full((N, 2 * h, 2 * w, Cout), float("nan"), device=dev)
    ops._conv_up2_taps(xd, pc, outt)
    assert maxdiff(outt.permute(0, 3, 1, 2), want) <= 5e-5
    got2 = ops.conv2d(xd, pc, up2=True, tile_cfg=7, split_k=1)   # forced direct
    assert maxdiff(got2.permute(0, 3, 1, 2), want) <= 2e-4
    if conv_math == "f16x3":
        # patch-resident plans: the upsample is applied while the input patch is staged (one low-resolution quad per 2 x 2 block of
        # the patch); must equal the materialised upsample + the same kernel, edges and odd sizes included
        for cfg in (13, 15) + ((14, 16) if Cout > 64 else ()) + (20, 21, 22):      # 20..22: the squarer tiles of round 6 (refused on narrow maps)
            try:
                got3 = ops.conv2d(xd, pc, up2=True, tile_cfg=cfg, split_k=1)
            except _lib.ArsegError:
                assert cfg >= 20 and 2 * w < 48, cfg
                continue
            assert maxdiff(got3.permute(0, 3, 1, 2), want) <= 2e-4, cfg
            mat = ops.conv2d(ops.resize_nhwc(xd, 2 * h, 2 * w, _lib.BILINEAR, False), pc, tile_cfg=cfg, split_k=1)
            assert maxdiff(got3, mat) <= 1e-5, cfg


def test_conv2d_f16x3_accuracy(dev):
    """The split-fp16 back end must stay at fp32-grade accuracy (not fp16-grade) on a deep, wide-dynamic-range GEMM:
    K = 4608 with activations spanning 1e-3..1e2; compared against an fp64 reference, next to the fp32 MFMA back end."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    g = np.random.Generator(np.random.PCG64(91))
    x = t((g.standard_normal((1, 512, 12, 20)) * np.exp(g.uniform(-7, 4.6, (1, 512, 12, 20)))).astype(np.float32))
    w = rnd(92, 128, 512, 3, 3, scale=0.02) * t(np.exp(g.uniform(-3, 3, (128, 1, 1, 1))).astype(np.float32))
    pc = PackedConv(w, None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, dev)
    want = F.conv2d(x.double(), w.double(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    err = {}
    for m in ("f32", "f16x3"):
        prev = ops.set_conv_math(m)
        try:
            got = ops.conv2d(xd, pc, tile_cfg=5, split_k=1).permute(0, 3, 1, 2).cpu().double()
        finally:
            ops.set_conv_math(prev)
        err[m] = float((got - want).abs().max() / want.abs().max())
    assert err["f32"] < 5e-6, err
    assert err["f16x3"] < 5e-6 and err["f16x3"] < 2 * err["f32"], err      # measured 1.5e-6 vs 2.3e-6; plain fp16 would be ~1e-3


def test_conv2d_f16x3_range_boundaries(dev):
    """The documented operating range of the split-fp16 back end (include/arseg_hip.h, ARSEG_MATH_F16X3), at its boundaries, against
    an fp64 reference: (a) activations that are ALL tiny (1e-7 .. 1e-3, fp16-subnormal hi / lo halves): absolute error per product
    <= 6e-8 |w|; (b) activations up to 1.3e5 (hi saturates at 65504, lo carries the rest): still exact to 22 / 11 bits; (c) beyond
    131008 the pair clamps: the result is the convolution of the clamped input, not garbage."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    g = np.random.Generator(np.random.PCG64(93))
    w = rnd(94, 64, 64, 3, 3, scale=0.05)
    pc = PackedConv(w, None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, dev)

    def run(x):
        prev = ops.set_conv_math("f16x3")
        try:
            # a direct plan, pinned: the range statement is about the split operand format.  (The autotuner may pick the Winograd route
            # for a shape; its transformed operands are up to ~100x the activations, see below.)
            return ops.conv2d(x.permute(0, 2, 3, 1).contiguous().to(dev), pc, tile_cfg=7, split_k=1).permute(0, 3, 1, 2).cpu().double()
        finally:
            ops.set_conv_math(prev)

    sign = t(np.where(g.uniform(size=(1, 64, 10, 12)) < 0.5, -1.0, 1.0).astype(np.float32))
    # (a) all tiny
    xa = sign * t(np.exp(g.uniform(np.log(1e-7), np.log(1e-3), (1, 64, 10, 12))).astype(np.float32))
    ea = (run(xa) - F.conv2d(xa.double(), w.double(), padding=1)).abs().max()
    bound_a = 6e-8 * float(w.abs().sum(dim=(1, 2, 3)).max())                       # sum_k |w_k| * 2^-24 per output
    assert float(ea) <= bound_a, (float(ea), bound_a)
    # (b) large: 1e3 .. 1.3e5 (a quarter of the values above 65504)
    xb = sign * t(np.exp(g.uniform(np.log(1e3), np.log(1.3e5), (1, 64, 10, 12))).astype(np.float32))
    wb = F.conv2d(xb.double(), w.double(), padding=1)
    eb = float((run(xb) - wb).abs().max() / wb.abs().max())
    assert eb <= 5e-4, eb                                                           # 11-bit lo above 65504 (2^-12 relative); ~1e-6 below
    xb2 = xb.clamp(-6.5e4, 6.5e4)
    wb2 = F.conv2d(xb2.double(), w.double(), padding=1)
    assert float((run(xb2) - wb2).abs().max() / wb2.abs().max()) <= 5e-6           # full precision up to 65504
    # (c) beyond the range: a documented clamp at +-131008
    xc = xb * 4.0
    wc = F.conv2d(xc.clamp(-131008.0, 131008.0).double(), w.double(), padding=1)
    assert float((run(xc) - wc).abs().max() / wc.abs().max()) <= 5e-4
    # Winograd route under f16x3: the input transform B^T d B amplifies by up to 100 (typically ~10); the transformed operands are
    # stored scaled by 2^-4 (exact, undone in the output transform), which puts the worst-case limit at |x| ~ 2e4 (include/arseg_hip.h).
    # Activations up to 1e4 keep fp32-grade accuracy, and O(1) data does not feel the raised subnormal floor.
    for lo_, hi_ in ((1e-2, 1e4), (1e-3, 1.0)):
        xw = sign * t(np.exp(g.uniform(np.log(lo_), np.log(hi_), (1, 64, 10, 12))).astype(np.float32))
        ww = F.conv2d(xw.double(), w.double(), padding=1)
        prev = ops.set_conv_math("f16x3")
        try:
            ow = torch.empty((1, 10, 12, 64), device=dev)
            ops._conv_wino(xw.permute(0, 2, 3, 1).contiguous().to(dev), pc, None, ow, 1, 10, 12)
        finally:
            ops.set_conv_math(prev)
        assert float((ow.permute(0, 3, 1, 2).cpu().double() - ww).abs().max() / ww.abs().max()) <= 5e-5, (lo_, hi_)


def test_conv2d_range_guard(dev, monkeypatch):
    """Operand range of the split-fp16 back end.  Input magnitudes up to 1e6 (beyond the 131008 clamp of the direct plans, far beyond the
    Winograd route's range): the f16x3 result is the conv of the clamped input -- and the kernels say so through the sticky device word
    (default mode: no host sync per conv; one read by ops.range_tripped()), after which the fp32 back end gives the fp32-grade result.
    In-range inputs never set the word.  The round-2 validation mode (ARSEG_CONV_RANGE_GUARD=host) still falls back per layer."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    g = np.random.Generator(np.random.PCG64(97))
    w = rnd(98, 64, 64, 3, 3, scale=0.05)
    pc = PackedConv(w, None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, dev)
    x = t((g.standard_normal((1, 64, 12, 20)) * np.exp(g.uniform(np.log(1e2), np.log(1e6), (1, 64, 12, 20)))).astype(np.float32))
    want = F.conv2d(x.double(), w.double(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    rel = lambda y, ref: float((y.permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max())
    prev = ops.set_conv_math("f16x3")
    try:
        assert ops._RANGE_MODE == "device"
        ops.range_tripped()                                     # clear
        small = ops.conv2d(xd * 1e-4, pc)                      # inside the range
        assert not ops.range_tripped()
        e_off = rel(ops.conv2d(xd, pc), want)
        assert ops.range_tripped() and not ops.range_tripped()      # set by the kernels, cleared by the read
        ops.set_conv_math("f32")
        e_f32 = rel(ops.conv2d(xd, pc), want)
        assert not ops.range_tripped()
        ops.set_conv_math("f16x3")
        for cfg in (1, 13):                                    # an implicit-GEMM plan and a patch-resident plan, explicitly
            ops.conv2d(xd, pc, tile_cfg=cfg, split_k=1)
            assert ops.range_tripped(), cfg
            ops.conv2d(xd * 1e-4, pc, tile_cfg=cfg, split_k=1)
            assert not ops.range_tripped(), cfg
        monkeypatch.setattr(ops._config.sw, "RANGE_GUARD", True)          # the host-synchronising validation mode
        e_host = rel(ops.conv2d(xd, pc), want)
        assert ops._math == _lib.MATH_F16X3
    finally:
        ops.set_conv_math(prev)
        ops.range_tripped()
    assert e_f32 <= 1e-5 and e_host <= 1e-5 and e_off > 1e-2, (e_f32, e_host, e_off)
    assert rel(small, want * 1e-4) <= 1e-5


def test_range_guard_winograd_and_evaluator(dev):
    """The Winograd route multiplies transformed activations (up to ~100x the input, stored scaled by 2^-4): its batched GEMM watches
    those (the evaluator's fallback: tests/test_gpu_models.py::test_evaluator_range_fallback)."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    w = rnd(198, 64, 64, 3, 3, scale=0.05)
    pc = PackedConv(w, None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, dev)
    if getattr(pc, "wino_u", None) is None:
        pytest.skip("no Winograd weights for this layer")
    x = rnd(199, 1, 24, 32, 64).to(dev)
    out = torch.empty((1, 24, 32, 64), dtype=torch.float32, device=dev)
    prev = ops.set_conv_math("f16x3")
    try:
        ops.range_tripped()
        ops._conv_wino(x, pc, None, out, 1, 24, 32)
        assert not ops.range_tripped()
        ops._conv_wino(x * 3e5, pc, None, out, 1, 24, 32)       # |V| * 2^-4 beyond 65504 for some tiles
        assert ops.range_tripped()
    finally:
        ops.set_conv_math(prev)


def test_range_guard_psp_pyramid_operand(dev):
    """ The per-image pyramid operand of the folded PSP bottleneck is un-scaled by 1 / scale[co] before it is split: a channel with a
    small folded scale can push it beyond 65504.  arseg_psp_w2_split_fwd carries the range watch of that operand (ABI level and through
    ops.psp_bottleneck_x3), as every other producer of split rows does."""
    import ctypes

    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda a: ctypes.c_void_p(a.data_ptr() if a is not None else None)      # noqa: E731
    N, rows, Cout = 2, 50, 64
    tt = rnd(700, N, rows, Cout).to(dev)
    un = torch.ones(Cout, device=dev)
    out = torch.empty((N, Cout, 64), dtype=torch.float32, device=dev)
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.arseg_psp_w2_split_fwd(P(tt), P(un), P(out), N, rows, Cout, P(word), 65504.0, st), "psp_w2_split")
    assert int(word.item()) == 0
    got = ops.SplitRows(out.view(N, Cout, 1, 64)).float().reshape(N, Cout, 64)
    assert maxdiff(got[:, :, :rows], tt.permute(0, 2, 1)) <= 2e-6 * float(tt.abs().max()) and float(got[:, :, rows:].abs().max()) == 0.0
    un[7] = 1e6                                                # one channel whose folded scale is tiny
    _lib.check(lib.arseg_psp_w2_split_fwd(P(tt), P(un), P(out), N, rows, Cout, P(word), 65504.0, st), "psp_w2_split")
    assert int(word.item()) == 1
    _lib.check(lib.arseg_psp_w2_split_fwd(P(tt), P(un), P(out), N, rows, Cout, None, 0.0, st), "psp_w2_split")      # no watch: allowed
    assert lib.arseg_psp_w2_split_fwd(P(tt), P(un), P(out), N, rows, Cout, ctypes.c_void_p(word.data_ptr() + 1), 65504.0, st) == _lib.ARSEG_EINVAL
    # through the host layer: a bottleneck whose channel 5 has a scale of 3e-4 x the others (still foldable) and pyramid terms of ~1e3
    C, H, W, sizes = 64, 12, 16, (1, 2, 3, 6)
    w = rnd(701, 128, C, scale=0.1)
    gamma = torch.ones(128); gamma[5] = 3e-4
    pc = PackedConv(w, None, (gamma, torch.zeros(128), torch.zeros(128), torch.ones(128)), act=_lib.ACT_RELU, device=dev)
    assert ops.psp_x3_foldable(pc)
    feats = rnd(702, 1, H, W, C).to(dev)
    prev = ops.configure(conv_math="f16x3", conv_range_guard="device", conv_gemm_x3=True)
    try:
        ops.range_tripped()
        ops.psp_bottleneck_x3(feats, (rnd(703, 1, 50, 128) * 1e-2).to(dev), pc, sizes)      # (the un-scale factor also holds the weights' power-of-two pre-scale)
        assert not ops.range_tripped()
        ops.psp_bottleneck_x3(feats, (rnd(703, 1, 50, 128) * 1e3).to(dev), pc, sizes)          # 1e3 / 3e-4 > 65504 in channel 5
        assert ops.range_tripped()
    finally:
        ops.configure(**prev)
        ops.range_tripped()


@pytest.mark.parametrize("H,W,h,w", [(36, 48, 18, 24), (35, 47, 17, 23), (20, 30, 20, 30)])
def test_frame_u8_ingest(dev, H, W, h, w):
    """uint8 HWC -> normalised NHWC4 in one kernel == ToTensor + Normalize + F.interpolate(align_corners=True) of the oracle,
    and == the two-step GPU path (float frame -> frame_to_nhwc4)."""
    from arseg_amd import ingest, ops
    from oracle import cpu_ref

    g = np.random.Generator(np.random.PCG64(61))
    img = g.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    got = ingest.frames_to_nhwc4(img, h, w, ingest.CITY_BISE_MEAN, ingest.CITY_BISE_STD, device=dev)
    norm = cpu_ref.to_tensor_normalize(img, ingest.CITY_BISE_MEAN, ingest.CITY_BISE_STD)
    want = F.interpolate(norm, (h, w), mode="bilinear", align_corners=True) if (h, w) != (H, W) else norm
    assert got.shape == (2, h, w, 4) and float(got[..., 3].abs().max()) == 0.0
    assert maxdiff(got[..., :3].permute(0, 3, 1, 2), want) <= 1e-5      # fp32 lerp of values up to +-4 (FMA contraction on the GPU)
    two_step = ops.frame_to_nhwc4(norm.to(dev), h, w)
    assert maxdiff(got, two_step) <= 1e-5


@pytest.mark.parametrize("H,W,F_", [(720, 960, 4), (37, 53, 11), (8, 8, 1)])
def test_merge_motion(dev, H, W, F_):
    """GPU mergeMotion == the oracle (itself pinned to the reference's function, G9), bit for bit, as the int16 arrays the
    dataset's .bin files hold (astype(np.short) of the oracle's int32)."""
    from arseg_amd import ops, synth
    from oracle import cpu_ref

    flows = synth.make_mv_chain(21 + F_, H, W, F_)
    want = cpu_ref.merge_motion(flows).transpose(2, 0, 1, 3).astype(np.int16)           # [F+1,H,W,2]
    got = ops.merge_motion(torch.from_numpy(flows).to(dev)).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


def test_conv2d_f16_reduced_precision(dev):
    """ARSEG_MATH_F16 (plain fp16 operands, fp32 accumulate): works on every tile shape and is, as documented, a reduced
    precision mode -- ~1e-3 relative, three orders of magnitude above the split-fp16 default."""
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    x = rnd(95, 2, 128, 20, 24)
    w = rnd(96, 96, 128, 3, 3, scale=0.03)
    pc = PackedConv(w, None, None, 1, 1, 1, _lib.ACT_NONE, 0.0, dev)
    want = F.conv2d(x.double(), w.double(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    prev = ops.set_conv_math("f16")
    try:
        errs = []
        for cfg in (5, 6, 7, 8, 9, 11, 1):
            got = ops.conv2d(xd, pc, tile_cfg=cfg, split_k=1).permute(0, 3, 1, 2).cpu().double()
            errs.append(float((got - want).abs().max() / want.abs().max()))
        got_w = torch.empty((2, 20, 24, 96), device=dev)
        ops._conv_wino(xd, pc, None, got_w, 2, 20, 24)
        errs.append(float((got_w.permute(0, 3, 1, 2).cpu().double() - want).abs().max() / want.abs().max()))
    finally:
        ops.set_conv_math(prev)
    assert max(errs[:-1]) < 2e-3 and min(errs) > 2e-5 and errs[-1] < 2e-2, errs      # direct ~5e-4; Winograd amplifies it to ~9e-3


def test_hw_probe_transpose_read_and_mfma_layout(dev, tmp_path):
    """Builds and runs ar-seg_amd/csrc/probes/probe_tr.hip: the LDS transpose-read lane semantics and the 16x16x32 fp16 MFMA
    operand layout that creff_mfma.hip is written against, checked on the actual GPU."""
    import os
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ar-seg_amd", "csrc", "probes", "probe_tr.hip")
    exe = str(tmp_path / "probe_tr")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", src, "-o", exe], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("N,H,W,C", [(2, 9, 13, 64), (1, 1, 1, 8), (1, 1, 7, 4), (1, 6, 1, 12), (1, 32, 64, 64)])
def test_upsample2x_fast_path(dev, N, H, W, C):
    """The dedicated x2 bilinear (align_corners=False) kernel behind resize_nhwc, incl. degenerate sizes and a strided
    (channel-slice) destination."""
    from arseg_amd import _lib, ops

    x = rnd(97, N, C, H, W)
    want = F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.resize_nhwc(xd, 2 * H, 2 * W, _lib.BILINEAR, False)
    assert maxdiff(got.permute(0, 3, 1, 2), want) <= 1e-6
    big = torch.full((N, 2 * H, 2 * W, C + 8), 3.0, device=dev)
    ops.resize_nhwc(xd, 2 * H, 2 * W, _lib.BILINEAR, False, out=big[..., 4:4 + C])
    assert maxdiff(big[..., 4:4 + C].permute(0, 3, 1, 2), want) <= 1e-6
    assert float(big[..., :4].min()) == 3.0 and float(big[..., 4 + C:].max()) == 3.0


# ---------------------------------------------------------------------------------------------- ops.creff_warp at C = 128 .. 512, fp32 / 16-bit inputs
@pytest.mark.parametrize("C,Hp,Wp,hp,wp,H,W,n_cls,logsm,layout,dtype", [
    (128, 33, 47, 17, 24, 33, 47, 19, False, "c8", torch.float32),          # identity MV resize, odd sizes
    (256, 24, 40, 12, 20, 192, 320, 19, False, "c8", torch.float32),        # BiSeNet: MVs at 8x the feature resolution
    (256, 21, 37, 10, 18, 168, 296, 19, False, "nhwc", torch.bfloat16),     # 16-bit inputs, non-integer lr ratio (configs[4] style)
    (256, 24, 40, 12, 20, 192, 320, 12, True, "c8", torch.float16),
    (512, 17, 20, 9, 10, 136, 160, 19, False, "c8", torch.float32),         # Cityscapes PSPNet
    (64, 20, 30, 10, 15, 20, 30, 12, True, "nhwc", torch.float16),          # 64 channels with 16-bit inputs (the fused kernel is fp32 only)
    (256, 24, 40, 12, 20, 192, 320, 19, False, "c8", torch.float16),
    (256, 24, 40, 12, 20, 192, 320, 12, False, "c8", torch.float32),
    (256, 24, 40, 12, 20, 192, 320, 12, True, "c8", torch.bfloat16),
])
def test_creff_warp_wide_and_16bit(dev, C, Hp, Wp, hp, wp, H, W, n_cls, logsm, layout, dtype):
    """ops.creff_warp (MV resize + warp + CReFF + head) beyond the 64-channel fp32 case of the fused kernel -- BiSeNet C = 256, Cityscapes
    PSPNet C = 512, fp16 / bf16 inputs, MVs at 8x the feature resolution, non-integer lr ratios -- against the oracle's MV resize -> warp ->
    MyAttention -> head evaluated on the same (rounded) inputs.  These shapes run as a warp launch + the matrix-core CReFF kernel."""
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention
    from oracle import cpu_ref

    N = 2
    g = np.random.Generator(np.random.PCG64(131))
    m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 9, attn_gain=0.35)
    sd = {kk: v.clone() for kk, v in m.state_dict().items()}
    refs = [rnd(140 + i, C, Hp, Wp).to(dtype) for i in range(N)]
    lr = rnd(142, N, C, hp, wp).to(dtype)
    s8 = H // Hp
    mvq = torch.from_numpy((g.integers(-3 * s8, 3 * s8 + 1, (N, H, W, 2)) * 4).astype(np.int16))
    mvq[1, : H // 2] = mvq[1, 0, 0]
    hr_w = torch.cat([cpu_ref.warp_feature(refs[i][None].float(), cpu_ref.mv_resize(cpu_ref.mv_from_int16(mvq[i:i + 1]), Hp, Wp)) for i in range(N)])
    want = cpu_ref.my_attention(sd, "", hr_w, lr.float(), 7, 7)
    pa = PackedAttention(m, dev)
    wf, bf = rnd(122, n_cls, C, scale=0.1), rnd(123, n_cls, scale=0.1)
    refs_d = [r.permute(1, 2, 0).contiguous().to(dev) for r in refs]
    lay = _lib.C8 if layout == "c8" else _lib.NHWC
    p, logits = ops.creff_warp(refs_d, mvq.to(dev), lr.permute(0, 2, 3, 1).contiguous().to(dev), pa, (wf.to(dev), bf.to(dev)), logsm, 7, 7, p_layout=lay)
    assert p.dtype == torch.float32 and logits.dtype == torch.float32
    got = ops.from_c8(p, _lib.NCHW) if layout == "c8" else p.permute(0, 3, 1, 2)
    tol = 2e-4
    assert maxdiff(got, want) <= tol
    lg = F.conv2d(want, wf[:, :, None, None], bf)
    if logsm:
        lg = F.log_softmax(lg, dim=1)
    assert maxdiff(logits, lg) <= 3 * tol


# ---------------------------------------------------------------------------------------------- warp + CReFF fused (C = 64)
@pytest.mark.parametrize("Hp,Wp,hp,wp,n_cls,logsm,layout", [
    (40, 70, 20, 35, 12, True, "c8"),       # ragged vs the 16x16 tile
    (16, 16, 8, 8, 12, True, "nhwc"),       # exactly one tile
    (33, 47, 17, 24, 19, False, "c8"),      # odd sizes, two classifier row blocks
    (7, 9, 7, 9, 0, False, "nhwc"),         # image smaller than a tile, same-size lr, no head
    (64, 96, 32, 48, 12, True, "c8"),       # several tiles in both directions
])
# ("roll", 0, 0): the rolling kernel with its default schedule (balanced runs of steps); ("roll", 6, 3): fixed 6-row segments on 3 workgroups
# (many pieces each); ("roll", 0, 5) / ("roll", 0, 11): the balanced schedule on few workgroups -- whole-strip passes plus a remainder cut mid-strip
@pytest.mark.parametrize("impl,seg_rows,max_wgs", [("roll", 0, 0), ("tiles", 0, 0), ("roll", 6, 3), ("roll", 0, 5), ("roll", 0, 11)])
def test_creff_warp_fused(dev, Hp, Wp, hp, wp, n_cls, logsm, layout, impl, seg_rows, max_wgs):
    """arseg_creff_warp_fwd_ex (MV warp fused into CReFF) against the oracle's warp -> MyAttention -> head: the rolling kernel
    (csrc/creff_roll.hip, the default; also with 6-row strip segments on 3 persistent workgroups, so that every workgroup walks several
    segments and every segment boundary lies inside the image) and the 16 x 16 tile kernel (csrc/creff_rr.hip)."""
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention
    from oracle import cpu_ref

    C, N = 64, 3
    g = np.random.Generator(np.random.PCG64(31))
    for gain in (0.35, 1.0):
        m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=gain)
        sd = {kk: v.clone() for kk, v in m.state_dict().items()}
        refs = [rnd(40, C, Hp, Wp), rnd(41, C, Hp, Wp)]
        refs = [refs[0], refs[1], refs[0]]                      # frames 0 and 2 share a keyframe
        lr = rnd(42, N, C, hp, wp)
        mvq = torch.from_numpy((g.integers(-9, 10, (N, Hp, Wp, 2)) * 4).astype(np.int16))
        mvq[1, : Hp // 2] = mvq[1, 0, 0]                        # a block-constant region
        mvq[2, :, : Wp // 3, 0] = 4 * (Wp + 5)                  # samples far outside the image -> zeros
        hr_w = torch.cat([cpu_ref.warp_feature(refs[i][None], cpu_ref.mv_resize(cpu_ref.mv_from_int16(mvq[i:i + 1]), Hp, Wp)) for i in range(N)])
        want = cpu_ref.my_attention(sd, "", hr_w, lr, 7, 7)
        pa = PackedAttention(m, dev)
        head = None
        if n_cls:
            wf, bf = rnd(22, n_cls, C, scale=0.2), rnd(23, n_cls, scale=0.1)
            head = (wf.to(dev), bf.to(dev))
        refs_d = [r.permute(1, 2, 0).contiguous().to(dev) for r in refs]
        lay = _lib.C8 if layout == "c8" else _lib.NHWC
        prev = ops.configure(creff_warp_impl=impl, creff_seg_rows=seg_rows, creff_max_wgs=max_wgs)
        try:
            if impl == "roll" and n_cls > 16:
                # one dispatch rule (tests/test_host_config.py::test_creff_dispatch_table): the rolling kernel has no 17-32-class head; pinned to it
                # such a launch is refused, under AUTO it runs on the tile kernel
                with pytest.raises(_lib.ArsegError):
                    ops.creff_warp(refs_d, mvq.to(dev), ops.to_nhwc(lr.to(dev)), pa, head, logsm, 7, 7, p_layout=lay)
                ops.configure(creff_warp_impl="")
                assert ops.creff_warp_kernel(N, C, Hp, Wp, hp, wp, n_cls) == "tiles"
            p, logits = ops.creff_warp(refs_d, mvq.to(dev), ops.to_nhwc(lr.to(dev)), pa, head, logsm, 7, 7, p_layout=lay)
        finally:
            ops.configure(**prev)
        got = ops.from_c8(p, _lib.NCHW) if layout == "c8" else p.permute(0, 3, 1, 2)
        tol = 1e-4 if gain < 1 else 3e-4
        assert maxdiff(got, want) <= tol
        if n_cls:
            lg = F.conv2d(want, wf[:, :, None, None], bf)
            if logsm:
                lg = F.log_softmax(lg, dim=1)
            assert maxdiff(logits, lg) <= 2 * tol
        else:
            assert logits is None


@pytest.mark.parametrize("B,M,K,K2,N", [(3, 300, 96, 32, 36), (11, 2048, 512, 64, 1024), (1, 77, 32, 64, 64)])
def test_gemm_x3_cat(dev, B, M, K, K2, N):
    """arseg_gemm_x3_cat_fwd: out = act(scale (x.w^T + x2.w2^T) + bias) with a batch-strided x / shared w and a shared x2 / batch-strided w2 (the
    folded PSP bottleneck's operand pattern), every tile_cfg, against fp64."""
    import ctypes

    from arseg_amd import _lib

    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda a: ctypes.c_void_p(a.data_ptr() if a is not None else None)      # noqa: E731
    x, w = rnd(500, B, M, K).to(dev), rnd(501, N, K, scale=0.1).to(dev)
    x2, w2 = rnd(502, M, K2).abs().to(dev), rnd(503, B, N, K2, scale=0.3).to(dev)
    scale, bias = rnd(504, N).abs().add(0.5).to(dev), rnd(505, N).to(dev)
    sp = {}
    for name, a in (("x", x), ("w", w), ("x2", x2), ("w2", w2)):
        sp[name] = torch.empty_like(a)
        _lib.check(lib.arseg_split_rows_fwd(P(a), a.shape[-1], P(sp[name]), a.numel() // a.shape[-1], a.shape[-1], 1.0, None, 0.0, st), "split")
    want = torch.relu((torch.einsum("bmk,nk->bmn", x.double(), w.double()) + torch.einsum("mk,bnk->bmn", x2.double(), w2.double())) * scale.double() + bias.double())
    for cfg in range(7):
        out = torch.full((B, M, N), float("nan"), device=dev)
        _lib.check(lib.arseg_gemm_x3_cat_fwd(P(sp["x"]), P(sp["w"]), P(sp["x2"]), P(sp["w2"]), P(out), M, N, K, K2, N, B, M * K * 4, 0, 0, N * K2 * 4, M * N,
                                             P(scale), P(bias), _lib.ACT_RELU, 0.0, 0, cfg, None, 0.0, st), "gemm_x3_cat")
        assert float((out.double() - want).abs().max()) <= 3e-6 * float(want.abs().max()), cfg
    assert lib.arseg_gemm_x3_cat_fwd(P(sp["x"]), P(sp["w"]), None, None, P(out), M, N, K, K2, N, B, 0, 0, 0, 0, 0, None, None, 0, 0.0, 0, 0, None, 0.0, st) == _lib.ARSEG_EINVAL


@pytest.mark.parametrize("independent", [False, True])
def test_gop_graph_lanes(dev, independent):
    """executor.GopGraph: lanes captured into one joined graph or into one graph per lane on its own stream; replays reproduce the eager result
    bit for bit and pick up inputs refilled in place."""
    from arseg_amd import _lib, ops
    from arseg_amd.executor import GopGraph
    from arseg_amd.packing import PackedConv

    x = rnd(900, 2, 24, 40, 64).to(dev)
    w = rnd(901, 64, 64, 3, 3, scale=0.05)
    pc1 = PackedConv(w, None, None, 1, 1, 1, _lib.ACT_RELU, 0.0, dev)
    pc2 = PackedConv(rnd(902, 32, 64, 1, 1, scale=0.1), None, None, 1, 0, 1, _lib.ACT_NONE, 0.0, dev)

    def step():
        return ops.conv2d(ops.conv2d(x, pc1), pc2)

    want = step().clone()
    g = GopGraph([step] * 3, warmup=1, independent=independent)
    for _ in range(2):
        outs = g.replay()
        g.synchronize()
        torch.cuda.synchronize()
        assert len(outs) == 3 and all(torch.equal(o, want) for o in outs)
    x.mul_(2.0)                                  # inputs are read from fixed tensors: refill in place, replay
    want2 = step().clone()
    outs = g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(o, want2) for o in outs) and not torch.equal(want2, want)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["nhwc", "c8"])
def test_creff_warp_batch_above_2gib_one_launch(dev, layout):
    """A batch whose fused feature exceeds 2 GiB (11 frames of 768 x 1024 x 64 fp32 = 2.2 GB; PSPNet at 1024 x 2048 is the bench's case) runs as
    ONE launch of the rolling kernel -- every frame has its own buffer descriptor, the 32-bit offsets span a frame -- and equals, bit for bit,
    the same frames launched one at a time.  Pinned to the tile kernel (one descriptor per tensor) the batch is split instead: same values
    within the two kernels' rounding."""
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention

    C, N, Hp, Wp, hp, wp, n_cls = 64, 11, 768, 1024, 384, 512, 12
    assert N * C * Hp * Wp * 4 >= (1 << 31)
    lay = _lib.C8 if layout == "c8" else _lib.NHWC
    gen = torch.Generator(device="cpu").manual_seed(5)
    m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=0.35)
    pa = PackedAttention(m, dev)
    refs = [torch.randn((Hp, Wp, C), generator=gen).to(dev) for _ in range(2)]
    refs_d = [refs[i % 2] for i in range(N)]
    lr = torch.randn((N, hp, wp, C), generator=gen).to(dev)
    mvq = (torch.randint(-9, 10, (N, Hp, Wp, 2), generator=gen) * 4).to(torch.int16).to(dev)
    head = (rnd(22, n_cls, C, scale=0.2).to(dev), rnd(23, n_cls, scale=0.1).to(dev))
    assert ops.creff_warp_kernel(N, C, Hp, Wp, hp, wp, n_cls) == "roll"
    with ops.profile() as prof:
        p, logits = ops.creff_warp(refs_d, mvq, lr, pa, head, True, 7, 7, p_layout=lay)
    assert prof.summary()["creff_warp"]["launches"] == 1
    assert tuple(logits.shape) == (N, n_cls, Hp, Wp)
    for i in (0, 5, 10):                        # first, middle, last frame: the offsets beyond 2 GiB are the last frames'
        p1, l1 = ops.creff_warp(refs_d[i:i + 1], mvq[i:i + 1], lr[i:i + 1], pa, head, True, 7, 7, p_layout=lay)
        assert torch.equal(p[i:i + 1], p1) and torch.equal(logits[i:i + 1], l1), i
    prev = ops.configure(creff_warp_impl="tiles")
    try:
        with ops.profile() as prof:
            pt, lt = ops.creff_warp(refs_d, mvq, lr, pa, head, True, 7, 7, p_layout=lay)
        assert prof.summary()["creff_warp"]["launches"] == 2          # 10 frames fit below 2 GiB, then 1
    finally:
        ops.configure(**prev)
    assert maxdiff(pt[-1:], p[-1:]) <= 2e-4 and maxdiff(lt[-1:], logits[-1:]) <= 4e-4
    assert maxdiff(pt[:1], p[:1]) <= 2e-4


@pytest.mark.gpu
def test_creff_wide_batch_above_2gib_slices(dev):
    """ops.creff (C >= 128 kernels: 32-bit offsets over a launch's batch) on a batch above 2 GiB: split into launches that write their slices of
    ONE output (no concatenation pass), equal bit for bit to the frames launched alone."""
    from arseg_amd import ops, synth
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention

    C, N, Hp, Wp, hp, wp, n_cls = 128, 5, 1024, 1024, 512, 512, 19
    assert N * C * Hp * Wp * 4 >= (1 << 31)
    gen = torch.Generator(device="cpu").manual_seed(6)
    pa = PackedAttention(synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=0.35), dev)
    hr = torch.randn((N, C // 8, Hp, Wp, 8), generator=gen).to(dev)
    lr = torch.randn((N, hp, wp, C), generator=gen).to(dev)
    head = (rnd(22, n_cls, C, scale=0.2).to(dev), rnd(23, n_cls, scale=0.1).to(dev))
    with ops.profile() as prof:
        p, logits = ops.creff(hr, lr, pa, head, True, 7, 7)
    assert prof.summary()["creff"]["launches"] == 2 and tuple(p.shape) == tuple(hr.shape) and tuple(logits.shape) == (N, n_cls, Hp, Wp)
    for i in (0, 3, 4):
        p1, l1 = ops.creff(hr[i:i + 1], lr[i:i + 1], pa, head, True, 7, 7)
        assert torch.equal(p[i:i + 1], p1) and torch.equal(logits[i:i + 1], l1), i


@pytest.mark.gpu
def test_creff_roll_schedules_vs_oracle_random(dev):
    """The rolling kernel on random small shapes (odd sizes, lr of any smaller size, 1-4 frames) under random schedules -- the balanced
    default and fixed segments, few and many workgroups -- against the oracle's warp -> MyAttention: every piece list must cover every
    pixel exactly once whatever the cut."""
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention
    from oracle import cpu_ref

    C = 64
    fz = np.random.Generator(np.random.PCG64(77))
    m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=0.35)
    sd = {kk: v.clone() for kk, v in m.state_dict().items()}
    pa = PackedAttention(m, dev)
    for case in range(24):
        Hp, Wp = int(fz.integers(2, 60)), int(fz.integers(2, 70))
        hp, wp, N = int(fz.integers(1, Hp + 1)), int(fz.integers(1, Wp + 1)), int(fz.integers(1, 5))
        seg_rows, max_wgs = int(fz.choice([0, 0, 2, 6, 14, 40])), int(fz.choice([0, 0, 1, 3, 7, 20]))
        refs = [rnd(100 + case * 8 + i, C, Hp, Wp) for i in range(N)]
        lr = rnd(300 + case, N, C, hp, wp)
        mvq = torch.from_numpy((fz.integers(-7, 8, (N, Hp, Wp, 2)) * 4 + fz.integers(0, 4, (N, Hp, Wp, 2))).astype(np.int16))
        hr_w = torch.cat([cpu_ref.warp_feature(refs[i][None], cpu_ref.mv_resize(cpu_ref.mv_from_int16(mvq[i:i + 1]), Hp, Wp)) for i in range(N)])
        want = cpu_ref.my_attention(sd, "", hr_w, lr, 7, 7)
        refs_d = [r.permute(1, 2, 0).contiguous().to(dev) for r in refs]
        prev = ops.configure(creff_warp_impl="roll", creff_seg_rows=seg_rows, creff_max_wgs=max_wgs)
        try:
            p, _ = ops.creff_warp(refs_d, mvq.to(dev), ops.to_nhwc(lr.to(dev)), pa, None, False, 7, 7, p_layout=_lib.NHWC)
        finally:
            ops.configure(**prev)
        assert maxdiff(p.permute(0, 3, 1, 2), want) <= 1e-4, (case, Hp, Wp, hp, wp, N, seg_rows, max_wgs)
