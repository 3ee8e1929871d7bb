#!/usr/bin/env python3
"""Per-phase shader-clock stamps of creff_mfma_kernel (C >= 128: BiSeNet / semseg) -- dev builds with -DMFMA_TIMING only:

    SRC=creff_mfma bash tools/build_rr_variant.sh mfma_timing -DMFMA_TIMING
    python tools/time_mfma.py scratch/rr_libs/lib_mfma_timing.so [--json out.json] [--ty 8|16]

Every wave adds the ticks between consecutive stamps to its row; the table shows wave 0, the mean and the slowest wave per TILE (16 channel chunks x 2
passes for C = 256).  The launch is BASELINE configs[2]'s CReFF stage: 11 frames, C = 256, 128 x 256 feature, lr 64 x 128, 19 classes."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ARSEG_HIP_LIB"] = os.path.abspath(sys.argv[1])
from arseg_amd import _lib, ops, synth  # noqa: E402
from arseg_amd.model import MyAttention  # noqa: E402
from arseg_amd.packing import PackedAttention  # noqa: E402

NAMES = ["p1 wait dma", "p1 barrier A", "p1 issue next", "p1 lr_up tile", "p1 key conv", "p1 barrier B", "p1 query conv", "p1 QK mfma", "softmax",
         "p2 wait dma", "p2 barrier A", "p2 issue + value conv + lr residual", "p2 barrier B", "p2 PV mfma + store + head", "logits"]
lib = _lib.load()
dev = torch.device("cuda:0")
B, C, Hp, Wp = 11, 256, 128, 256
if "--ty" in sys.argv:
    ops.configure(creff_tile_rows=int(sys.argv[sys.argv.index("--ty") + 1]))
g = np.random.Generator(np.random.PCG64(5))
hr = ops.to_c8(torch.from_numpy(g.standard_normal((B, Hp, Wp, C)).astype(np.float32)).to(dev), _lib.NHWC)
lr = torch.from_numpy(g.standard_normal((B, Hp // 2, Wp // 2, C)).astype(np.float32)).to(dev)
pa = PackedAttention(synth.load_synth_weights(MyAttention(C, kW=7, kH=7), 7, attn_gain=0.35), dev)
wf = torch.from_numpy((0.2 * g.standard_normal((19, C))).astype(np.float32)).to(dev)
bf = torch.from_numpy((0.1 * g.standard_normal(19)).astype(np.float32)).to(dev)
rd = lib.arseg__mfma_dbg_read
rd.restype, rd.argtypes = None, [ctypes.c_void_p, ctypes.c_int]
for _ in range(3):
    ops.creff(hr, lr, pa, (wf, bf), False)
rd(None, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); ops.creff(hr, lr, pa, (wf, bf), False); e.record(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
rd(buf, 0)
d = np.array(list(buf), dtype=np.float64).reshape(16, 16)
n = d[0, 15]
nw = int((d[:, 0] > 0).sum())
print("launch ms", s.elapsed_time(e), "tiles", int(n), "waves per workgroup", nw)
tot = d[0, :15].sum()
res = {"launch_ms": s.elapsed_time(e), "tiles": int(n), "phases": {}}
print(f"{'phase':38s} {'wave0':>8s} {'mean':>8s} {'max':>8s}   share(wave0)")
for i, nm in enumerate(NAMES):
    col = d[:nw, i] / n
    res["phases"][nm] = {"wave0": col[0], "mean": col.mean(), "max": col.max()}
    print(f"{nm:38s} {col[0]:8.0f} {col.mean():8.0f} {col.max():8.0f}   {100 * d[0, i] / tot:5.1f}%")
print("total ticks per tile (wave 0)", tot / n)
res["total_ticks_per_tile"] = tot / n
if "--json" in sys.argv:
    json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
