#!/usr/bin/env python3
"""Times the patch-resident 3x3 conv plans (tile_cfg 13..16) on the 64- / 128-channel layer shapes of the headline network (GPU only;
ARSEG_HIP_LIB selects a variant build of the library).

    python tools/bench_patch.py [--reps 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from arseg_amd import _lib, ops  # noqa: E402
from arseg_amd.packing import PackedConv  # noqa: E402

SHAPES = [  # name, N, H, W (conv input = output size), Cin, Cout, up2
    ("layer1 LR", 11, 64, 128, 64, 64, False),
    ("layer2 LR", 11, 32, 64, 128, 128, False),
    ("up_3 LR", 11, 256, 512, 64, 64, True),
    ("layer1 HR", 1, 128, 256, 64, 64, False),
    ("layer2 HR", 1, 64, 128, 128, 128, False),
    ("up_3 HR", 1, 512, 1024, 64, 64, True),
]


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
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    _lib.load()
    dev = torch.device("cuda:0")
    ops.set_conv_math("f16x3")
    g = np.random.Generator(np.random.PCG64(5))
    for name, N, H, W, Cin, Cout, up2 in SHAPES:
        h, w = (H // 2, W // 2) if up2 else (H, W)
        x = torch.from_numpy(g.standard_normal((N, h, w, Cin)).astype(np.float32)).to(dev)
        wt = torch.from_numpy((g.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32))
        pc = PackedConv(wt, None, None, 1, 1, 1, _lib.ACT_RELU, 0.0, dev)
        ref = None
        if not args.no_check:
            xin = x.permute(0, 3, 1, 2)
            if up2:
                xin = F.interpolate(xin, scale_factor=2.0, mode="bilinear", align_corners=False)
            ref = F.relu(F.conv2d(xin, wt.to(dev), padding=1)).permute(0, 2, 3, 1)
        line = f"{name:10s}"
        for cfg in (13, 14, 15, 16):
            try:
                out = ops.conv2d(x, pc, tile_cfg=cfg, split_k=1, up2=up2)
            except _lib.ArsegError:
                line += f"  cfg{cfg}    n/a       "
                continue
            err = float((out - ref).abs().max()) if ref is not None else float("nan")
            us = timeit(lambda: ops.conv2d(x, pc, tile_cfg=cfg, split_k=1, up2=up2, out=out), args.reps)
            line += f"  cfg{cfg} {us:7.1f} us ({err:.0e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
