#!/usr/bin/env python3
"""Per-phase shader-clock stamps of creff_rr_kernel (dev builds with -DRR_TIMING only; the shipped library has no stamps).

    hipcc ... -DRR_TIMING -c ar-seg_amd/csrc/creff_rr.hip ; link into a copy of the library (scratch/build_var.sh)
    python tools/time_phases.py <lib.so> [--json out.json]

Every wave adds the ticks between consecutive stamps to its own row; the table shows wave 0, the mean over the 16 waves and
the slowest wave of each phase, per tile.  (s_memtime instrumentation itself costs ~10 %.)
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ARSEG_HIP_LIB"] = os.path.abspath(sys.argv[1])
from arseg_amd import _lib, ops, synth  # noqa: E402
from arseg_amd.model import MyAttention  # noqa: E402
from arseg_amd.packing import PackedAttention  # noqa: E402

ORDER = [(12, "taps"), (9, "window->lds"), (1, "lr_up build"), (2, "qconv"), (3, "gather rest"), (4, "h-load"), (5, "kconv"), (10, "qk-mfma"),
         (6, "softmax"), (7, "vconv+wfs"), (14, "pv: mfma blocks"), (0, "pv: epilogue 0-1"), (11, "pv: epilogue 2-3"), (13, "logits"), (8, "end barrier")]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    H, W, B, C = 512, 1024, 11, 64
    g = np.random.Generator(np.random.PCG64(5))
    clip = synth.make_clip(0, H, W, gop=B + 1, mean=synth.CAMVID_MEAN, std=synth.CAMVID_STD)
    mvq = torch.from_numpy(clip["mv"][1:B + 1]).to(dev)
    ref = torch.from_numpy(g.standard_normal((H, W, C)).astype(np.float32)).to(dev)
    lr = torch.from_numpy(g.standard_normal((B, H // 2, W // 2, C)).astype(np.float32)).to(dev)
    m = synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=0.35)
    pa = PackedAttention(m, dev)
    wf = torch.from_numpy((0.2 * g.standard_normal((12, C))).astype(np.float32)).to(dev)
    bf = torch.from_numpy((0.1 * g.standard_normal(12)).astype(np.float32)).to(dev)
    dbg = torch.zeros(16 * 16, dtype=torch.int64, device=dev)
    fn = lib.arseg__rr_set_dbg
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p]
    for _ in range(3):
        ops.creff_warp([ref] * B, mvq, lr, pa, (wf, bf), True)
    torch.cuda.synchronize()
    fn(dbg.data_ptr())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.creff_warp([ref] * B, mvq, lr, pa, (wf, bf), True)
    e.record()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(16, 16).astype(np.float64)
    n = d[0, 15]
    print("launch ms", s.elapsed_time(e), "tiles", int(n))
    tot = sum(d[0, i] for i, _ in ORDER)
    res = {"launch_ms": s.elapsed_time(e), "tiles": int(n), "phases": {}}
    print(f"{'phase':18s} {'wave0':>8s} {'mean':>8s} {'max':>8s}   share(wave0)")
    for i, nm in ORDER:
        col = d[:, i] / n
        res["phases"][nm] = {"wave0": col[0], "mean": col.mean(), "max": col.max()}
        print(f"{nm:18s} {col[0]:8.0f} {col.mean():8.0f} {col.max():8.0f}   {100 * d[0, i] / tot:5.1f}%")
    print("total ticks/tile (wave 0)", tot / n)
    res["total_ticks_per_tile"] = tot / n
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
