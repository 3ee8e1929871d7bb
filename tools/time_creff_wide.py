#!/usr/bin/env python3
"""creff_mfma_kernel (C >= 128: BiSeNet 256 at 128x256, semseg 512) alone: us per launch and per frame for the bench shapes (11 frames, one frame)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arseg_amd import _lib, ops, synth
from arseg_amd.model import MyAttention
from arseg_amd.packing import PackedAttention
dev = torch.device("cuda:0")
g = np.random.Generator(np.random.PCG64(5))
for (B, C, Hp, Wp, ncls) in ((11, 256, 128, 256, 19), (11, 512, 128, 256, 19), (1, 256, 128, 256, 19)):
    hr = ops.to_c8(torch.from_numpy(g.standard_normal((B, Hp, Wp, C)).astype(np.float32)).to(dev), _lib.NHWC)
    lr = torch.from_numpy(g.standard_normal((B, Hp // 2, Wp // 2, C)).astype(np.float32)).to(dev)
    pa = PackedAttention(synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=0.35), dev)
    wf = torch.from_numpy((0.2 * g.standard_normal((ncls, C))).astype(np.float32)).to(dev)
    bf = torch.from_numpy((0.1 * g.standard_normal(ncls)).astype(np.float32)).to(dev)
    t = ops._time(lambda: ops.creff(hr, lr, pa, (wf, bf), False), reps=10, rounds=3)
    print(f"creff_mfma B={B} C={C} {Hp}x{Wp}: {1e3*t:.1f} us per launch, {1e3*t/B:.1f} us per frame")
