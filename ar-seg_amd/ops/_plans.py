"""Per-shape launch plans (tile shape, LDS buffering, split-K, route): the cache, its optional JSON mirror, and the timing helpers.

The library's built-in heuristic is good to ~10 %; the first time a conv shape is seen on a device the candidates are timed with HIP
events and the fastest is cached (what MIOpen calls "find").  ARSEG_CONV_AUTOTUNE=0 keeps the heuristic."""
from __future__ import annotations

import os

import torch

from .. import _lib
from . import _config
from ._config import sw


class _PlanCache(dict):
    """Plans keyed by shape tuples; optionally mirrored to a JSON file (``ops.configure(conv_plan_file=...)`` merges that file in)."""

    def __init__(self):
        super().__init__()
        self.load()

    def load(self):
        if sw.PLAN_FILE and os.path.exists(sw.PLAN_FILE):
            import json

            with open(sw.PLAN_FILE) as f:
                for k, v in json.load(f).items():
                    super().__setitem__(tuple(json.loads(k)), tuple(v) if isinstance(v, list) else v)

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        if sw.PLAN_FILE:
            import json

            with open(sw.PLAN_FILE, "w") as f:
                json.dump({json.dumps(list(k)): (list(v) if isinstance(v, tuple) else v) for k, v in self.items()}, f, indent=0)


_conv_plans = _PlanCache()
_config._plan_file_listeners.append(_conv_plans.load)
_PATCH_CFGS = (13, 14, 15, 16, 20, 21, 22)      # arseg_conv_desc.tile_cfg of the patch-resident 3x3 kernel (the only direct plans with a fused x2 upsample)


def _conv_candidates(ktiles: int, cout: int, m: int, patch_ok: bool = False):
    cands = []
    if patch_ok:                                   # patch-resident 3x3 kernel (13/14: 128-pixel tiles, 15/16: 256; BN 64/128)
        cands += [(15, 1), (13, 1), (20, 1), (21, 1), (22, 1)] + ([(16, 1), (14, 1)] if cout > 64 else [])      # 20..22: squarer 64-channel tiles (r6)
    for cfg in (5, 6, 7, 8, 9, 10, 11, 12) + ((17, 18, 19) if sw.math == _lib.MATH_F16X3 else ()):
        bn = {17: 128, 18: 256, 19: 256}.get(cfg, 128 if cfg in (5, 8, 9, 12) else 64)
        bm = {17: 256, 18: 128, 19: 256}.get(cfg, 128 if cfg in (5, 6, 9, 10) else 64)
        if bn >= 128 and cfg >= 17 and cout < bn:
            continue
        if bn == 128 and cout <= 64:
            continue
        if bm == 128 and m <= 64:
            continue
        for sk in (1, 2, 3, 4, 6, 8):
            if sk > 1 and (ktiles // sk < 4 or cout % 4):
                continue
            cands.append((cfg, sk))
    return cands


def _time(fn, reps=6, rounds=2):
    """ms per call: the faster of ``rounds`` averages over ``reps`` calls (plans chosen from one short average were visibly noisy box to box)."""
    fn()
    best = float("inf")
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best


def _tune_conv(launch, pc, m, allow_patch=True):
    ktiles = (pc.R * pc.S * pc.cin_pad + 31) // 32
    best, best_t = (0, 0), float("inf")
    patch_ok = allow_patch and (sw.math == _lib.MATH_F16X3 and pc.R == 3 and pc.S == 3 and pc.stride == 1 and pc.pad == pc.dil == 1 and pc.cin_pad % 32 == 0)
    for cfg, sk in ([(0, 0)] if allow_patch else []) + _conv_candidates(ktiles, pc.cout, m, patch_ok):
        try:
            launch(cfg, sk, record=False)                          # warm (also sizes the workspace)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                launch(cfg, sk, record=False)
            e.record()
            e.synchronize()
            t = s.elapsed_time(e)
        except _lib.ArsegError:
            continue
        if t < best_t:
            best, best_t = (cfg, sk), t
    return best if best_t < float("inf") else None          # None: no candidate could be launched (nothing to cache)
