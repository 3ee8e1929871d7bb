#!/usr/bin/env python3
"""Per-segment shader-clock stamps of creff_roll_kernel (dev builds with -DROLL_TIMING only; the shipped library has no stamps).

    SRC=creff_roll bash tools/build_rr_variant.sh timing -DROLL_TIMING
    python tools/time_roll.py scratch/rr_libs/lib_timing.so [--json out.json]

Every wave accumulates the ticks of its four segments per iteration -- H1 work, wait at barrier A, H2 work, wait at barrier B -- over the
launch; the table shows them per step (two query rows of a 16-column strip), per wave (waves 0-1: consumers, 2-7: producers).
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
    ops.configure(creff_warp_impl="roll")
    nw = 16
    dbg = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
    fn = lib.arseg__roll_set_dbg
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
    d = dbg.cpu().numpy().reshape(nw, 8).astype(np.float64)
    res = {"launch_ms": s.elapsed_time(e), "waves": []}
    print("launch ms", s.elapsed_time(e))
    print(f"{'wave':>4s} {'H1':>8s} {'waitA':>8s} {'H2':>8s} {'waitB':>8s} {'total':>8s}   s_memtime ticks per step (about 0.5 ns each: launch time / iterations)")
    for w in range(nw):
        n = d[w, 4]
        if n == 0:
            continue
        row = d[w, :4] / n
        res["waves"].append({"wave": w, "H1": row[0], "waitA": row[1], "H2": row[2], "waitB": row[3]})
        print(f"{w:4d} {row[0]:8.1f} {row[1]:8.1f} {row[2]:8.1f} {row[3]:8.1f} {row.sum():8.1f}")
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
