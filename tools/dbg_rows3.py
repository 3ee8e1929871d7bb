import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import torch.nn.functional as F
from arseg_amd import _lib, ops
from arseg_amd.packing import PackedConv
dev = torch.device("cuda:0")
def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))
N, H, W, Cin, Cout, dil = 2, 12, 20, 64, 64, 1
x = rnd(130, N, Cin, H, W)
w = rnd(132, Cout, Cin, 3, 3, scale=float(np.sqrt(2.0 / (Cin * 9))))
for use_bn in (False, True):
  for use_res in (False, True):
    g = np.random.Generator(np.random.PCG64(131))
    bn = None
    if use_bn:
        bn = (torch.from_numpy(g.uniform(0.5, 1.5, Cout).astype(np.float32)), rnd(134, Cout, scale=0.1), rnd(135, Cout, scale=0.1), torch.from_numpy(g.uniform(0.5, 1.5, Cout).astype(np.float32)))
    pc = PackedConv(w, None, bn, 1, dil, dil, _lib.ACT_NONE, 0.2, dev)
    res = rnd(136, N, Cout, H, W) if use_res else None
    y = F.conv2d(x.double(), w.double(), None, padding=dil, dilation=dil)
    if bn is not None:
        y = F.batch_norm(y, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5)
    if res is not None:
        y = y + res.double()
    xp = ops.pad_rows(x.permute(0, 2, 3, 1).contiguous().to(dev), dil)
    rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    ref2 = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().to(dev), pc, residual=rd)
    print("bn", use_bn, "res", use_res, "conv2d err", float((ref2.permute(0, 3, 1, 2).cpu().double() - y).abs().max()))
    for cfg in range(12):
        got = ops.conv3x3_rows(xp, pc, residual=rd, cfg=cfg)
        e = (got.permute(0, 3, 1, 2).cpu().double() - y).abs()
        print("  cfg", cfg, "err", float(e.max()), "bad px", int((e.amax(1) > 1e-3).sum()), "of", N * H * W)
