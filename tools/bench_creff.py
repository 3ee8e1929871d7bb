#!/usr/bin/env python3
"""Times the warp + CReFF stage alone at a BASELINE configuration (default: PSPNet, 512x1024, 11 non-keyframes of one GOP):
the two-kernel path (arseg_warp_mvq_fwd + arseg_creff_fwd) against the fused kernel (arseg_creff_warp_fwd; C = 64 fp32 only -- other
--C / --dtype time ops.creff_warp's warp launch + matrix-core CReFF), and reports
the stage's algorithmic HBM roofline fraction (329.3 MB per 512x1024 frame).

    python tools/bench_creff.py [--H 512 --W 1024 --frames 11 --iters 20]
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
    ap.add_argument("--H", type=int, default=512)
    ap.add_argument("--W", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=11)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--n-cls", type=int, default=12)
    ap.add_argument("--skip-old", action="store_true")
    ap.add_argument("--C", type=int, default=64, help="channels (a multiple of 64: 256 = BiSeNet, 512 = Cityscapes PSPNet)")
    ap.add_argument("--feat-div", type=int, default=1, help="feature resolution = frame / feat_div (BiSeNet / semseg: 8); --H / --W are the FEATURE size")
    ap.add_argument("--dtype", choices=["f32", "f16", "bf16"], default="f32", help="element type of ref / lr")
    args = ap.parse_args()
    from arseg_amd import _lib, ops, synth
    from arseg_amd.model import MyAttention
    from arseg_amd.packing import PackedAttention

    dev = torch.device("cuda:0")
    H, W, B, C = args.H, args.W, args.frames, args.C
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    g = np.random.Generator(np.random.PCG64(5))
    clip = synth.make_clip(0, H * args.feat_div, W * args.feat_div, gop=B + 1, mean=synth.CAMVID_MEAN, std=synth.CAMVID_STD)
    mvq = torch.from_numpy(clip["mv"][1:B + 1]).to(dev)
    ref = torch.from_numpy(g.standard_normal((H, W, C)).astype(np.float32)).to(dev).to(dt)
    lr = torch.from_numpy(g.standard_normal((B, H // 2, W // 2, C)).astype(np.float32)).to(dev).to(dt)
    m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=0.35)
    pa = PackedAttention(m, dev)
    wf = torch.from_numpy((0.2 * g.standard_normal((args.n_cls, C))).astype(np.float32)).to(dev)
    bf = torch.from_numpy((0.1 * g.standard_normal(args.n_cls)).astype(np.float32)).to(dev)
    head = (wf, bf)

    def old():          # the two-kernel path: warp launch(es) -> fp32 C8 tensor -> arseg_creff_fwd
        ref_c8 = torch.empty((B, C // 8, H, W, 8), dtype=torch.float32, device=dev)
        for b in range(B):
            ops.warp_mvq(ref.unsqueeze(0), mvq[b:b + 1], _lib.C8, out=ref_c8[b:b + 1])
        return ops.creff(ref_c8, lr, pa, head, True, 7, 7)

    if C != 64 or dt != torch.float32:      # only the 64-channel fp32 case has a fused kernel: ops.creff_warp IS the two-kernel path here
        args.skip_old = True

    def new():
        return ops.creff_warp([ref] * B, mvq, lr, pa, head, True, 7, 7)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / args.iters

    eb = 4 if dt == torch.float32 else 2
    stage_bytes = (eb + 4) * C * H * W + eb * C * (H // 2) * (W // 2) + 4 * H * W * args.feat_div ** 2 + 4 * args.n_cls * H * W
    res = {"feature": [H, W], "C": C, "dtype": args.dtype, "frames": B, "algorithmic_bytes_per_frame": stage_bytes}
    p_new, l_new = new()
    t_new = timeit(new)
    res["fused_ms_per_frame"] = t_new / B
    res["fused_frac_hbm"] = stage_bytes / (t_new / B * 1e-3) / 8e12
    if not args.skip_old:
        p_old, l_old = old()
        res["max_abs_diff_p"] = float((p_new - p_old).abs().max())
        res["max_abs_diff_logits"] = float((l_new - l_old).abs().max())
        t_old = timeit(old)
        res["two_kernel_ms_per_frame"] = t_old / B
        res["two_kernel_frac_hbm"] = stage_bytes / (t_old / B * 1e-3) / 8e12
    print(json.dumps(res))


if __name__ == "__main__":
    main()
