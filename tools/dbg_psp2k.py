#!/usr/bin/env python3
"""Why does the rolling CReFF kernel lose 13 % of its roofline fraction at 1024x2048?  The same launch -- 3 frames, C = 64, 12 classes,
2M pixels per frame -- as a WIDE map (1024 x 2048: a 16-pixel strip's rows are 512 KB apart) and as a TALL one (2048 x 1024: 256 KB apart), and the
headline map (512 x 1024 x 11 frames) for reference: us per frame and per megapixel."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arseg_amd import ops, synth
from arseg_amd.model import MyAttention
from arseg_amd.packing import PackedAttention

dev = torch.device("cuda:0")
g = np.random.Generator(np.random.PCG64(3))
m = synth.load_synth_weights(MyAttention(64, kW=7, kH=7), 7)
pa = PackedAttention(m, dev)
wf, bf = torch.randn(12, 64).mul(0.2).to(dev), torch.randn(12).mul(0.1).to(dev)
res = {}
for name, (B, Hp, Wp) in {"512x1024x11": (11, 512, 1024), "1024x2048x3": (3, 1024, 2048), "2048x1024x3": (3, 2048, 1024), "1024x2048x1": (1, 1024, 2048),
                          "512x4096x3": (3, 512, 4096), "4096x512x3": (3, 4096, 512)}.items():
    ref = torch.randn(Hp, Wp, 64, device=dev)
    lr = torch.randn(B, Hp // 2, Wp // 2, 64, device=dev)
    mv = torch.from_numpy((g.integers(-6, 7, (B, Hp // 16, Wp // 16, 1, 1, 2)) * 4).astype(np.int16)).expand(B, Hp // 16, Wp // 16, 16, 16, 2)
    mv = mv.permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, 2).contiguous().to(dev)
    fn = lambda: ops.creff_warp([ref] * B, mv, lr, pa, (wf, bf), True, 7, 7, p_layout=1)
    t = ops._time(fn, reps=5, rounds=3)
    res[name] = {"ms": t, "us_per_frame": 1e3 * t / B, "us_per_Mpx": 1e3 * t / (B * Hp * Wp / 2 ** 20), "kernel": ops.creff_warp_kernel(B, 64, Hp, Wp, Hp // 2, Wp // 2, 12)}
    print(name, res[name], flush=True)
    del ref, lr, mv
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
