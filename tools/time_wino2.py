#!/usr/bin/env python3
"""Per-phase shader-clock stamps of wino2_kernel (dev builds with -DW2_TIMING):  SRC=wino2 bash tools/build_rr_variant.sh w2_timing -DW2_TIMING ;
python tools/time_wino2.py scratch/rr_libs/lib_w2_timing.so [--up2]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ARSEG_HIP_LIB"] = os.path.abspath(sys.argv[1])
from arseg_amd import _lib, ops  # noqa: E402
from arseg_amd.packing import PackedConv  # noqa: E402
NAMES = ["prologue (weights, first patch)", "-", "-", "phase A transform + MFMA", "dma wait + barrier A", "phase B output transform + stores (+ up2 build)", "barrier B"]
lib = _lib.load(); dev = torch.device("cuda:0")
up2 = "--up2" in sys.argv
N, H, W = (11, 256, 512) if up2 else (11, 64, 128)
g = np.random.Generator(np.random.PCG64(1))
x = torch.from_numpy(g.standard_normal((N, H // 2 if up2 else H, W // 2 if up2 else W, 64)).astype(np.float32)).to(dev)
pc = PackedConv(torch.from_numpy((g.standard_normal((64, 64, 3, 3)) * 0.06).astype(np.float32)), None, None, 1, 1, 1, _lib.ACT_RELU, 0.0, dev)
rd = lib.arseg__w2_dbg_read; rd.restype, rd.argtypes = None, [ctypes.c_void_p, ctypes.c_int]
for _ in range(3): ops.conv3x3_wino2(x, pc, up2=up2)
rd(None, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); ops.conv3x3_wino2(x, pc, up2=up2); e.record(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)(); rd(buf, 0)
d = np.array(list(buf), dtype=np.float64).reshape(16, 8)
wgs = d[0, 7]; blocks = N * ((H + 7) // 8) * ((W + 7) // 8)
print("launch ms", s.elapsed_time(e), "workgroups", int(wgs), "blocks", blocks, "blocks per workgroup", blocks / wgs)
for i, nm in enumerate(NAMES):
    col = d[:, i] / blocks * (1 if i else blocks / wgs)
    print(f"{nm:40s} wave0 {col[0]:8.0f}  mean {col.mean():8.0f}  max {col.max():8.0f}   (ticks per block{' / per workgroup' if i == 0 else ''})")
print("ticks per block (wave 0, loop only)", d[0, 1:7].sum() / blocks)
