import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from arseg_amd import _lib, ops
from arseg_amd.packing import PackedConv
dev = torch.device("cuda:0")
def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))
for (N, H, W, up2) in ((1, 8, 8, False), (2, 13, 21, False), (1, 16, 24, True), (11, 64, 128, False), (11, 256, 512, True), (1, 512, 1024, True), (1, 128, 256, False)):
    h, w = (H // 2, W // 2) if up2 else (H, W)
    x = rnd(1, N, 64, h, w)
    wt = rnd(2, 64, 64, 3, 3, scale=float(np.sqrt(2.0 / 576)))
    g = np.random.Generator(np.random.PCG64(3))
    bn = (torch.from_numpy(g.uniform(0.5, 1.5, 64).astype(np.float32)), rnd(4, 64, scale=0.1), rnd(5, 64, scale=0.1), torch.from_numpy(g.uniform(0.5, 1.5, 64).astype(np.float32)))
    pc = PackedConv(wt, None, bn, 1, 1, 1, _lib.ACT_RELU, 0.0, dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    res = rnd(6, N, H, W, 64).to(dev) if not up2 else None
    got = ops.conv3x3_wino2(xd, pc, residual=res, up2=up2)
    prev = ops.configure(conv_igemm3=False)
    ref = ops.conv2d(xd, pc, residual=res, up2=up2)
    t_old = ops._time(lambda: ops.conv2d(xd, pc, residual=res, up2=up2), reps=10)
    ops.configure(**prev)
    err_k = float((got - ref).abs().max())
    if N * H * W <= 200000:
        xin = F.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=False) if up2 else x.double()
        y = F.conv2d(xin, wt.double(), padding=1)
        y = F.batch_norm(y, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5)
        if res is not None: y = y + res.cpu().double().permute(0, 3, 1, 2)
        y = F.relu(y)
        err = float((got.cpu().double().permute(0, 3, 1, 2) - y).abs().max()); err_old = float((ref.cpu().double().permute(0, 3, 1, 2) - y).abs().max())
    else:
        err = err_old = None
    t_new = ops._time(lambda: ops.conv3x3_wino2(xd, pc, residual=res, up2=up2, record=False), reps=10)
    print(f"N={N} {H}x{W} up2={up2}: err vs fp64 {err} (old kernel {err_old}), vs old kernel {err_k:.2e}; old {1e3*t_old:.1f} us, wino2 {1e3*t_new:.1f} us", flush=True)
