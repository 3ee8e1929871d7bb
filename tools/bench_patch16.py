#!/usr/bin/env python3
"""The patch-resident 16-bit 3x3 kernel (arseg_conv2d16_fwd, tile_cfg 5..8) on the 3x3 stride-1 layer shapes of the BiSeNet bench
configurations, every tile_cfg timed alone with HIP events (5..8: the 4 x 64 / 2 x 64 pixel tiles, 10..13: the squarer tiles of round 6; with
ARSEG_HIP_LIB pointing at a -DPATCH_ABL=<bits> build the ablations that say where the kernel's time goes):

    python tools/bench_patch16.py [--dtype bf16|f16] [--json FILE]

Per shape: us and TFLOP/s (direct-conv FLOPs) of each tile_cfg, best implicit-GEMM plan (tile_cfg 1..4) beside them."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

SHAPES = [(11, 128, 256, 64, 64), (11, 64, 128, 128, 128), (11, 32, 64, 256, 256), (11, 16, 32, 512, 512), (11, 64, 128, 256, 256), (11, 32, 64, 256, 128),
          (11, 64, 128, 128, 64), (11, 32, 64, 128, 128), (1, 256, 512, 64, 64), (1, 128, 256, 128, 128), (1, 64, 128, 256, 256), (1, 32, 64, 512, 512),
          (1, 128, 256, 256, 256), (11, 39, 77, 256, 256), (11, 77, 154, 64, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--json")
    ap.add_argument("--cfgs", default="1,2,3,4,5,6,7,8,10,11,12,13")
    ap.add_argument("--shapes", type=int, default=0, help="only the first n shapes")
    args = ap.parse_args()
    cfgs = tuple(int(c) for c in args.cfgs.split(","))
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    dev = torch.device("cuda:0")
    sdt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    g = np.random.Generator(np.random.PCG64(5))
    rows = []
    for (N, H, W, Cin, Cout) in (SHAPES[:args.shapes] if args.shapes else SHAPES):
        w = torch.from_numpy((g.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32))
        pc = PackedConv(w, None, (torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout), torch.ones(Cout)), 1, 1, 1, _lib.ACT_RELU, 0.0, dev)
        x = torch.from_numpy(g.standard_normal((N, H, W, Cin)).astype(np.float32)).to(dev).to(sdt)
        res = torch.from_numpy(g.standard_normal((N, H, W, Cout)).astype(np.float32)).to(dev).to(sdt)
        flops = 2.0 * N * H * W * Cout * 9 * Cin
        per = {}
        for cfg in cfgs:
            try:
                ops.conv2d(x, pc, residual=res, tile_cfg=cfg)
                per[cfg] = 1e3 * ops._time(lambda: ops.conv2d(x, pc, residual=res, tile_cfg=cfg), reps=20)
            except _lib.ArsegError:
                pass
        row = {"N": N, "H": H, "W": W, "cin": Cin, "cout": Cout, "gflop": flops / 1e9, "us_by_cfg": {str(k): v for k, v in per.items()},
               "best": min(per, key=per.get), "best_tflops": flops / min(per.values()) / 1e6}
        rows.append(row)
        print(f"{N}x{H}x{W} {Cin}->{Cout}: " + "  ".join(f"{k}:{v:.1f}" for k, v in per.items()) + f"   best {row['best']} {row['best_tflops']:.0f} TF/s", flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"note": __doc__, "dtype": args.dtype, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
