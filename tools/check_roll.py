#!/usr/bin/env python3
"""Development check of the rolling warp + CReFF kernel (csrc/creff_roll.hip) against the tile kernel (csrc/creff_rr.hip) on the GPU:
both evaluate the same arithmetic in the same order, so they agree to rounding of the last bit or exactly.  Prints one JSON line per
case {shape, max_abs_p, max_abs_logits, roll_ms, tiles_ms}.  The parity tests proper (vs the oracle and the reference's fixtures) are
in tests/test_gpu_ops.py.

    python tools/check_roll.py [--big] [--seg-rows n]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also the 512x1024 x 11-frame headline shape (timed)")
    ap.add_argument("--only-big", action="store_true")
    ap.add_argument("--seg-rows", type=int, default=0)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fuzz", type=int, default=0, help="n random small shapes with random schedules (seg_rows, max_wgs) instead of the fixed list")
    args = ap.parse_args()
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention

    dev = torch.device("cuda:0")
    C = 64
    m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=0.35)
    pa = PackedAttention(m, dev)
    ops.configure(creff_seg_rows=args.seg_rows)

    def run(impl, refs, mvq, lr, head, layout, seg_rows=None, max_wgs=0):
        ops.configure(creff_warp_impl=impl, creff_seg_rows=args.seg_rows if seg_rows is None else seg_rows, creff_max_wgs=max_wgs)
        return ops.creff_warp(refs, mvq, lr, pa, head, True, 7, 7, layout)

    def timeit(fn, iters):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    # (Hp, Wp, hp, wp, B, mv_div (MV map = feature * mv_div), n_cls, layout)
    cases = [(12, 16, 6, 8, 1, 1, 12, _lib.C8), (10, 12, 5, 6, 2, 1, 12, _lib.NHWC), (7, 9, 3, 4, 1, 1, 19, _lib.C8),
             (32, 48, 16, 24, 2, 1, 12, _lib.NHWC), (33, 50, 16, 24, 1, 1, 0, _lib.C8), (64, 96, 32, 48, 3, 2, 12, _lib.NHWC),
             (140, 40, 70, 20, 2, 1, 12, _lib.C8), (300, 64, 150, 32, 1, 1, 12, _lib.NHWC)]
    sched = {}
    if args.fuzz:
        fz = np.random.Generator(np.random.PCG64(2024))
        cases = []
        for i in range(args.fuzz):
            Hp, Wp = int(fz.integers(2, 90)), int(fz.integers(2, 110))
            hp, wp = int(fz.integers(1, Hp + 1)), int(fz.integers(1, Wp + 1))
            case = (Hp, Wp, hp, wp, int(fz.integers(1, 5)), 1, int(fz.choice([0, 5, 12, 16])), int(fz.choice([_lib.C8, _lib.NHWC])))
            cases.append(case)
            sched[len(cases) - 1] = (int(fz.choice([0, 0, 2, 6, 14, 40])), int(fz.choice([0, 0, 1, 3, 7, 20, 100])))
    if args.only_big:
        cases = []
    if args.big or args.only_big:
        cases.append((512, 1024, 256, 512, 11, 1, 12, _lib.C8))
    g = np.random.Generator(np.random.PCG64(11))
    ok = True
    for ci, (Hp, Wp, hp, wp, B, mvd, n_cls, layout) in enumerate(cases):
        H, W = Hp * mvd, Wp * mvd
        big = Hp >= 512
        if big:
            clip = synth.make_clip(0, H, W, gop=B + 1, mean=synth.CAMVID_MEAN, std=synth.CAMVID_STD)
            mvq = torch.from_numpy(clip["mv"][1:B + 1]).to(dev)
        else:
            mv = g.integers(-6 * mvd, 6 * mvd + 1, size=(B, (H + 7) // 8, (W + 7) // 8, 2)) * 4
            mv = np.repeat(np.repeat(mv, 8, axis=1), 8, axis=2)[:, :H, :W]
            mv[:, : H // 3] += g.integers(-2, 3, size=(B, H // 3, W, 2))          # a non-rigid, fractional part
            mvq = torch.from_numpy(mv.astype(np.int16)).to(dev)
        refs = [torch.from_numpy(g.standard_normal((Hp, Wp, C)).astype(np.float32)).to(dev) for _ in range(1 if big else B)]
        if big:
            refs = refs * B
        lr = torch.from_numpy(g.standard_normal((B, hp, wp, C)).astype(np.float32)).to(dev)
        head = None
        if n_cls:
            head = (torch.from_numpy((0.2 * g.standard_normal((n_cls, C))).astype(np.float32)).to(dev),
                    torch.from_numpy((0.1 * g.standard_normal(n_cls)).astype(np.float32)).to(dev))
        p_t, l_t = run("tiles", refs, mvq, lr, head, layout)
        sr, mw = sched.get(ci, (None, 0))
        p_r, l_r = run("roll", refs, mvq, lr, head, layout, sr, mw)
        torch.cuda.synchronize()
        res = {"shape": [Hp, Wp, hp, wp, B, mvd, n_cls], "seg_rows": sr, "max_wgs": mw, "max_abs_p": float((p_t - p_r).abs().max()),
               "max_abs_logits": float((l_t - l_r).abs().max()) if n_cls else None,
               "nan": bool(torch.isnan(p_r).any())}
        if res["max_abs_p"] > 2.5e-4 or res["nan"] or (n_cls and res["max_abs_logits"] > 5e-4):      # (both kernels are within 1e-4 of the oracle)
            ok = False
            d = (p_t - p_r).abs()
            if layout == _lib.NHWC:
                bad = (d.amax(dim=3) > 1e-4).nonzero()
            else:
                bad = (d.amax(dim=(1, 4)) > 1e-4).nonzero()
            res["n_bad_px"] = int(bad.shape[0])
            res["first_bad"] = bad[:6].tolist()
            res["last_bad"] = bad[-3:].tolist()
        if big:
            res["roll_ms_per_frame"] = timeit(lambda: run("roll", refs, mvq, lr, head, layout), args.iters) / B
            res["tiles_ms_per_frame"] = timeit(lambda: run("tiles", refs, mvq, lr, head, layout), args.iters) / B
            res["roll_frac_hbm"] = 329252864 / (res["roll_ms_per_frame"] * 1e-3) / 8e12
        print(json.dumps(res), flush=True)
    print("ALL_OK" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
