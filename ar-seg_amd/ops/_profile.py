"""Optional per-launch timing (bench.py): HIP events on the launch stream around every ABI call."""
from __future__ import annotations

import torch

from .._lib import check

current = None          # the active ``profile`` context, if any


class profile:
    """``with ops.profile() as prof: ...`` records (op name, algorithmic flops, algorithmic bytes, ms) per launch.
    Events are recorded on torch's current stream, which is the stream handed to the library.  ``only`` = a set of op names: time
    just those (the others launch without events, so that concurrent streams keep the GPU as busy as in an un-instrumented run)."""

    def __init__(self, only=None):
        self.only = None if only is None else frozenset(only)

    def __enter__(self):
        global current
        self.records = []
        current = self
        return self

    def __exit__(self, *exc):
        global current
        current = None
        torch.cuda.synchronize()
        self.rows = [(name, flops, nbytes, s.elapsed_time(e)) for name, flops, nbytes, s, e, _ in self.records]
        self.tags = [tag for *_, tag in self.records]
        return False

    def summary(self):
        out = {}
        for name, flops, nbytes, ms in self.rows:
            r = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0, "bytes": 0})
            r["launches"] += 1
            r["ms"] += ms
            r["flops"] += flops
            r["bytes"] += nbytes
        return out

    def layers(self):
        """Per conv layer (one row per distinct shape + plan): every launch made on behalf of the layer (GEMM / patch kernel, Winograd
        transforms, tap gather, materialised upsample), with the reference's direct-conv FLOP count of the layer."""
        out = {}
        for (name, flops, nbytes, ms), tag in zip(self.rows, self.tags):
            if tag is None:
                continue
            r = out.setdefault(tag, {"calls": 0, "ms": 0.0, "kernels": {}})
            r["ms"] += ms
            r["kernels"][name] = r["kernels"].get(name, 0.0) + ms
            r["calls"] += name == "conv2d"
        rows = []
        for (N, H, W, cin, cout, k, stride, dil, up2, plan, flops), r in out.items():
            calls = max(r["calls"], 1)
            rows.append({"N": N, "H": H, "W": W, "cin": cin, "cout": cout, "k": k, "stride": stride, "dil": dil, "up2": up2, "plan": plan,
                         "calls": calls, "us_per_call": 1e3 * r["ms"] / calls, "gflop_per_call": flops / 1e9,
                         "tflops": flops * calls / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else None,
                         "us_by_kernel": {kk: 1e3 * v / calls for kk, v in r["kernels"].items()}})
        return rows


layer_tag = None      # set by conv2d while it launches on behalf of one layer (profile.layers())


def launch(name, fn, *args, flops=0, nbytes=0):
    """Every ABI call of the host layer goes through here: status check, and HIP events around it while a profile is active."""
    if current is None or (current.only is not None and name not in current.only):
        check(fn(*args), name)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    check(fn(*args), name)
    e.record()
    current.records.append((name, flops, nbytes, s, e, layer_tag))


class tagged:
    """``with tagged(tag): ...`` -- launches made inside belong to one conv layer of profile.layers(); the outermost tag wins (the
    tap-decomposed route calls conv2d for its low-resolution GEMM).  ``tag`` may be a callable (built only while a profile is active)."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        global layer_tag
        self.outer = layer_tag
        if current is not None and self.outer is None:
            layer_tag = self.tag() if callable(self.tag) else self.tag
        return self

    def __exit__(self, *exc):
        global layer_tag
        layer_tag = self.outer
        return False
