#!/usr/bin/env python3
"""The implicit 3x3 GEMM of the LDS-DMA kernel (arseg_conv3x3_rows_fwd, csrc/gemm_x3.hip) alone, on the 3x3 stride-1 layer shapes of the
bench configurations, against the plan the tuner picks today for the same layer (ops.conv2d with conv_igemm3 off: implicit GEMM with register
staging / patch-resident kernel / Winograd on the fp32 path, conv16 kernels on the 16-bit path):

    python tools/bench_rows3.py [--json FILE] [--set psp|bise16|all]

Per shape: us of the best existing plan, of the pad pass (NHWC -> zero-bordered rows), of every tile_cfg of the new kernel with an NHWC output and
with a padded output (the form a conv -> conv chain uses: no pad pass, no un-padded round trip), TFLOP/s in the layer's direct-conv FLOPs."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

SHAPES = {
    # N, H, W, Cin, Cout, dil  (H, W = the conv's input = output size)
    "psp": [(11, 64, 128, 64, 64, 1), (11, 32, 64, 128, 128, 1), (11, 32, 64, 128, 256, 1), (11, 32, 64, 256, 256, 1), (11, 32, 64, 256, 256, 2),
            (11, 32, 64, 256, 512, 1), (11, 32, 64, 512, 512, 1), (11, 32, 64, 512, 512, 4), (11, 256, 512, 64, 64, 1),
            (1, 128, 256, 64, 64, 1), (1, 64, 128, 128, 128, 1), (1, 64, 128, 256, 256, 1), (1, 64, 128, 512, 512, 1), (1, 512, 1024, 64, 64, 1)],
    "bise16": [(11, 128, 256, 64, 64, 1), (11, 64, 128, 128, 128, 1), (11, 32, 64, 256, 256, 1), (11, 16, 32, 512, 512, 1), (11, 64, 128, 256, 256, 1),
               (11, 32, 64, 256, 128, 1), (11, 64, 128, 128, 64, 1), (1, 256, 512, 64, 64, 1), (1, 128, 256, 128, 128, 1), (1, 64, 128, 256, 256, 1),
               (1, 32, 64, 512, 512, 1), (1, 128, 256, 256, 256, 1)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--set", default="all", choices=["psp", "bise16", "all"])
    args = ap.parse_args()
    from arseg_amd import _lib, ops
    from arseg_amd.packing import PackedConv

    dev = torch.device("cuda:0")
    g = np.random.Generator(np.random.PCG64(5))
    rows = []
    for name in (("psp", "bise16") if args.set == "all" else (args.set,)):
        sdt = torch.float32 if name == "psp" else torch.bfloat16
        for (N, H, W, Cin, Cout, dil) in SHAPES[name]:
            w = torch.from_numpy((g.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32))
            pc = PackedConv(w, None, (torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout), torch.ones(Cout)), 1, dil, dil, _lib.ACT_RELU, 0.0, dev)
            x = torch.from_numpy(g.standard_normal((N, H, W, Cin)).astype(np.float32)).to(dev).to(sdt)
            flops = 2.0 * N * H * W * Cout * 9 * Cin
            prev = ops.configure(conv_igemm3=False)
            try:
                ops.conv2d(x, pc)                                      # tunes
                t_old = ops._time(lambda: ops.conv2d(x, pc), reps=10)
                plan = [v for k, v in ops._conv_plans.items() if (k[0] == "conv16" and k[3:8] == (N, H, W, Cin, Cout)) or
                        (len(k) == 13 and k[1:6] == (N, H, W, Cin, Cout) and k[10] == dil)]
            finally:
                ops.configure(**prev)
            t_pad = ops._time(lambda: ops.pad_rows(x, dil), reps=10)
            xp = ops.pad_rows(x, dil)
            per_cfg = {}
            for cfg in range(12):
                try:
                    a = ops._time(lambda: ops.conv3x3_rows(xp, pc, cfg=cfg, record=False), reps=10)
                    b = ops._time(lambda: ops.conv3x3_rows(xp, pc, out_padded=True, cfg=cfg, record=False), reps=10) if Cout % 32 == 0 else None
                    per_cfg[cfg] = (a, b)
                except _lib.ArsegError:
                    pass
            best = min(per_cfg, key=lambda c: per_cfg[c][0])
            bestp = min((c for c in per_cfg if per_cfg[c][1] is not None), key=lambda c: per_cfg[c][1], default=None)
            row = {"set": name, "N": N, "H": H, "W": W, "cin": Cin, "cout": Cout, "dil": dil, "gflop": flops / 1e9, "old_plan": str(plan[-1]) if plan else None,
                   "old_us": 1e3 * t_old, "pad_us": 1e3 * t_pad, "rows_nhwc_us": 1e3 * per_cfg[best][0], "rows_nhwc_cfg": best,
                   "rows_padded_us": 1e3 * per_cfg[bestp][1] if bestp is not None else None, "rows_padded_cfg": bestp,
                   "old_tflops": flops / t_old / 1e9, "rows_padded_tflops": flops / per_cfg[bestp][1] / 1e9 if bestp is not None else None,
                   "us_by_cfg": {str(c): [1e3 * v[0], None if v[1] is None else 1e3 * v[1]] for c, v in per_cfg.items()}}
            rows.append(row)
            print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items() if k != "us_by_cfg"}, flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"note": __doc__, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
