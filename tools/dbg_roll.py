#!/usr/bin/env python3
"""Debug: the failing case of test_creff_warp_fused for both impls vs the oracle; prints where the error is."""
import os, sys, json
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from arseg_amd import _lib, ops, synth
from arseg_amd.model import MyAttention
from arseg_amd.packing import PackedAttention
from oracle import cpu_ref

def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))

dev = torch.device("cuda:0")
Hp, Wp, hp, wp, n_cls = 40, 70, 20, 35, 12
C, N = 64, 3
g = np.random.Generator(np.random.PCG64(31))
import itertools
for gain, variant in itertools.product((1.0,), ("base", "mv0")):
    m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=gain)
    sd = {kk: v.clone() for kk, v in m.state_dict().items()}
    refs = [rnd(40, C, Hp, Wp), rnd(41, C, Hp, Wp)]
    refs = [refs[0], refs[1], refs[0]]
    lr = rnd(42, N, C, hp, wp)
    mvq = torch.from_numpy((g.integers(-9, 10, (N, Hp, Wp, 2)) * 4).astype(np.int16))
    mvq[1, : Hp // 2] = mvq[1, 0, 0]
    mvq[2, :, : Wp // 3, 0] = 4 * (Wp + 5)
    if variant == "mv0":
        mvq[:] = 0
    if variant == "mvblock":
        mvq[:] = mvq[:, :1, :1]
    if variant == "refconst":
        refs = [r * 0 + r[:, :1, :1] for r in refs]
    if variant == "lrconst":
        lr = lr * 0 + lr[:, :, :1, :1]
    hr_w = torch.cat([cpu_ref.warp_feature(refs[i][None], cpu_ref.mv_resize(cpu_ref.mv_from_int16(mvq[i:i + 1]), Hp, Wp)) for i in range(N)])
    want = cpu_ref.my_attention(sd, "", hr_w, lr, 7, 7)
    pa = PackedAttention(m, dev)
    refs_d = [r.permute(1, 2, 0).contiguous().to(dev) for r in refs]
    for impl in ("tiles", "roll"):
        ops.configure(creff_warp_impl=impl)
        p, _ = ops.creff_warp(refs_d, mvq.to(dev), ops.to_nhwc(lr.to(dev)), pa, None, False, 7, 7, p_layout=_lib.NHWC)
        got = p.permute(0, 3, 1, 2).cpu()
        if impl == "roll":
            reps = [ops.creff_warp(refs_d, mvq.to(dev), ops.to_nhwc(lr.to(dev)), pa, None, False, 7, 7, p_layout=_lib.NHWC)[0].permute(0, 3, 1, 2).cpu() for _ in range(4)]
            print("   run-to-run max diff:", [float((r - got).abs().max()) for r in reps])
        d = (got - want).abs().amax(dim=1)          # [N, Hp, Wp]
        print(gain, variant, impl, "max", float(d.max()), "per-frame", [float(d[i].max()) for i in range(N)], "mean", float(d.mean()))
        idx = (d > 0.5 * d.max()).nonzero()
        print("   worst px:", idx[:10].tolist(), "|want| max", float(want.abs().max()))
        rows = d.amax(dim=(0, 2)); cols = d.amax(dim=(0, 1))
        if impl == "roll":
            print("   by row :", " ".join(f"{v*1e4:.1f}" for v in rows.tolist()))
            print("   by col :", " ".join(f"{v*1e4:.1f}" for v in cols.tolist()))
