#!/usr/bin/env python3
"""A/B of the two f16x3 GEMM kernels on the GEMM shapes of the headline network (GPU only).

    python tools/bench_gemm_x3.py [--json out.json] [--reps 20]

arseg_gemm_x3_fwd (operands pre-split in HBM, LDS-DMA staged, csrc/gemm_x3.hip) against arseg_conv2d_fwd in batched 1x1 mode
(fp32 activations split on the way into LDS, csrc/conv_igemm.hip) with its best tile_cfg, same weights, same data; both are checked
against an fp64 matmul."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from arseg_amd import _lib  # noqa: E402
from arseg_amd._lib import ConvDesc  # noqa: E402

SHAPES = [  # name, batch, M, K, N
    ("wino 512->512 (layer4)", 36, 1408, 512, 512),
    ("wino 256->512", 36, 1408, 256, 512),
    ("wino 256->256 (layer3)", 36, 1408, 256, 256),
    ("wino 128->256", 36, 1408, 128, 256),
    ("up_1 taps 1024->9x256", 1, 22528, 1024, 2304),
    ("up_2 taps 256->9x64", 1, 90112, 256, 576),
    ("psp 1x1 512->1024", 1, 22528, 512, 1024),
    ("HR wino 512->512", 36, 512, 512, 512),
    ("HR up_1 taps", 1, 8192, 1024, 2304),
]


def ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else None)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--cfgs", default="0,1,2,3,4,5,6", help="tile_cfg values of the new kernel (0-11).  Ablation builds: compile csrc/gemm_x3.hip with -DARSEG_GX3_ABLATE and pass 16 * a + c (a = 1: no DMA, 2: no fragment reads, 3: neither, 4: raised MFMA priority; wrong results, same instruction stream otherwise)")
    args = ap.parse_args()
    global CFGS
    CFGS = [int(c) for c in args.cfgs.split(",")]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = []
    for si, (name, B, M, K, N) in enumerate(SHAPES):
        if args.only >= 0 and si != args.only:
            continue
        g = torch.Generator(device="cpu").manual_seed(1)
        x = (torch.randn(B, M, K, generator=g) * torch.exp(torch.randn(B, M, 1, generator=g))).to(dev)
        w = (torch.randn(B, N, K, generator=g) * 0.05).to(dev)
        xs = torch.empty(B, M, K, dtype=torch.float32, device=dev)      # split rows occupy the same 4 bytes per value
        ws = torch.empty(B, N, K, dtype=torch.float32, device=dev)
        _lib.check(lib.arseg_split_rows_fwd(ptr(x), K, ptr(xs), B * M, K, 1.0, None, 0.0, st), "split x")
        _lib.check(lib.arseg_split_rows_fwd(ptr(w), K, ptr(ws), B * N, K, 1.0, None, 0.0, st), "split w")
        out = torch.empty(B, M, N, dtype=torch.float32, device=dev)
        ref = torch.bmm(x.double(), w.double().transpose(1, 2))
        rmax = ref.abs().max().item()
        rec = {"shape": name, "batch": B, "M": M, "K": K, "N": N, "gflop": 2.0 * B * M * K * N / 1e9}
        # new kernel, every tile_cfg
        best = None
        for cfg in CFGS:
            out.zero_()

            def run(cfg=cfg):
                _lib.check(lib.arseg_gemm_x3_fwd(ptr(xs), ptr(ws), ptr(out), M, N, K, N, B, M * K * 4, N * K * 4, M * N, None, None, None, 0, 0, 0.0, 0, cfg, None, 0.0, st), "gemm_x3")

            run()
            torch.cuda.synchronize()
            err = (out.double() - ref).abs().max().item() / rmax
            us = timeit(run, args.reps)
            rec[f"x3_cfg{cfg}_us"] = us
            rec[f"x3_cfg{cfg}_err"] = err
            if best is None or us < best[0]:
                best = (us, cfg)
        rec["x3_us"], rec["x3_cfg"] = best
        # the implicit-GEMM kernel in batched 1x1 mode
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin, d.in_ld = 1, M, 1, K, K
        d.Cout, d.out_ld, d.res_ld = N, N, N
        d.R, d.S, d.stride, d.pad, d.dil = 1, 1, 1, 0, 1
        d.act, d.prelu_slope = _lib.ACT_NONE, 0.0
        d.batch, d.in_batch_stride, d.w_batch_stride, d.out_batch_stride = B, M * K, N * K, M * N
        d.math = _lib.MATH_F16X3
        bo = None
        for cfg in (0, 5, 6, 7, 8, 9, 10, 11, 12, 17, 18, 19):
            d.tile_cfg, d.split_k = cfg, 1

            def run_old():
                _lib.check(lib.arseg_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(ws), None, None, None, ptr(out), None, 0, st), "conv2d")

            try:
                out.zero_()
                run_old()
                torch.cuda.synchronize()
            except _lib.ArsegError:
                continue
            err = (out.double() - ref).abs().max().item() / rmax
            us = timeit(run_old, args.reps)
            if bo is None or us < bo[0]:
                bo = (us, cfg, err)
        rec["igemm_us"], rec["igemm_cfg"], rec["igemm_err"] = bo
        rec["speedup"] = bo[0] / best[0]
        rec["x3_pflops_executed"] = 3 * rec["gflop"] * 1e9 / (best[0] * 1e-6) / 1e15
        rec["igemm_pflops_executed"] = 3 * rec["gflop"] * 1e9 / (bo[0] * 1e-6) / 1e15
        rows.append(rec)
        print(f"{name:28s} x3 {best[0]:8.1f} us (cfg {best[1]}, err {rec['x3_cfg%d_err' % best[1]]:.1e}, {rec['x3_pflops_executed']:.2f} PF/s executed)   "
              f"igemm {bo[0]:8.1f} us (cfg {bo[1]}, err {bo[2]:.1e}, {rec['igemm_pflops_executed']:.2f})   x{rec['speedup']:.2f}   "
              + " ".join(f"{rec['x3_cfg%d_us' % c]:.0f}" for c in CFGS), flush=True)
    if args.json:
        json.dump({"note": __doc__, "rows": rows}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
