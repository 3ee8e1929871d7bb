#!/usr/bin/env python3
"""Which torch (aten) operators a GOP step of a bench configuration executes beside the library's own launches -- every one of them is a
kernel or a copy the HIP path did not ask for by name (GPU only):

    python tools/trace_aten.py [--config psp|psp2k|bise_bf16|bise03_fp16]

Prints, per aten operator that touches device memory, the call count per step and the arseg_amd source line it comes from."""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

VIEW_OPS = ("view", "reshape", "permute", "slice", "select", "unsqueeze", "squeeze", "expand", "as_strided", "transpose", "t.", "alias", "detach",
            "_unsafe_view", "unbind", "split", "narrow", "unfold", "sym_", "size", "stride", "is_", "numel", "dim", "_local_scalar_dense", "lift_fresh")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW_OPS):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "ar-seg_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            self.rows[(name, where, str(shapes))] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="psp")
    a = ap.parse_args()
    import bench
    from arseg_amd import evaluation as ev, ops, synth
    from arseg_amd.gop import GopRunner

    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS[a.config]
    H, W, SCALE = cfg["H"], cfg["W"], cfg.get("scale", 0.5)
    hr, lr, _, _ = bench.build_nets(dev, cfg)
    storage = cfg.get("storage", "f32")
    if storage != "f32":
        sdt = {"bf16": torch.bfloat16, "f16": torch.float16}[storage]
        hr.set_storage(sdt)
        lr.set_storage(sdt)
    mean, std = (synth.CAMVID_MEAN, synth.CAMVID_STD) if cfg["kind"] == "psp" else (synth.CITY_BISE_MEAN, synth.CITY_BISE_STD)
    clip = synth.make_clip(0, H, W, gop=12, mean=mean, std=std)
    key = torch.from_numpy(clip["frames"][0:1]).to(dev)
    fb = torch.from_numpy(clip["frames"][1:12]).to(dev)
    mb = torch.from_numpy(clip["mv"][1:12]).to(dev)
    fused_tail = cfg["kind"] == "bise"

    def step():
        with torch.no_grad():
            ref = ops.to_nhwc(hr(key)[-1])[0]
            if fused_tail:
                return ev.alter_res_batch_pred(lr, [ref] * 11, fb, mb, SCALE)[0]
            return ev.alter_res_batch_fast(lr, [ref] * 11, fb, mb, SCALE)[0]

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    log = Log()
    with log:
        step()
    torch.cuda.synchronize()
    for (name, where, shapes), n in sorted(log.rows.items(), key=lambda kv: -kv[1]):
        print(f"{n:4d}  {name:40s} {where:28s} {shapes}")


if __name__ == "__main__":
    main()
