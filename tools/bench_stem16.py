#!/usr/bin/env python3
"""The 16-bit stem kernel (conv16_stem_kernel, tile_cfg 9) alone on the two bench shapes (11 x 512 x 1024 LR batch, 1 x 1024 x 2048 keyframe), us per call.
With ARSEG_HIP_LIB pointing at a -DSTEM_ABL=<bits> build (tools/build_rr_variant.sh, SRC=conv16): the ablations that say where its time goes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from arseg_amd import _lib, ops
from arseg_amd.packing import PackedConv

dev = torch.device("cuda:0")
torch.manual_seed(0)
w = torch.randn(64, 3, 7, 7) * 0.1
res = {}
for name, (N, H, W) in {"lr": (11, 512, 1024), "hr": (1, 1024, 2048)}.items():
    for dt in (torch.bfloat16,):
        x = torch.zeros(N, H, W, 8, dtype=dt, device=dev)
        x[..., :3] = torch.randn(N, H, W, 3, device=dev).to(dt)
        pc = PackedConv(w, None, None, 2, 3, 1, _lib.ACT_RELU, 0.0, dev)
        res[name] = 1e3 * ops._time(lambda: ops.conv2d(x, pc, tile_cfg=9), reps=20)
        y = ops.conv2d(x, pc, tile_cfg=9)
        res[name + "_maxpool"] = 1e3 * ops._time(lambda: ops.maxpool3x3s2(y), reps=20)
print(json.dumps({"lib": os.environ.get("ARSEG_HIP_LIB", "default"), "us": res}))
