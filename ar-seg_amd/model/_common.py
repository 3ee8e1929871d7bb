"""Shared plumbing of the model mirrors: lazy weight packing + the inference-only guard."""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib


class HipModule(nn.Module):
    """nn.Module whose forward runs on libarseg_hip.so.

    Parameters / buffers live in ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` children so that
    ``state_dict()`` has exactly the reference's keys; the kernel-side layouts (``_pack``) are
    built on first use and dropped whenever the parameters may have changed (``load_state_dict``,
    ``.cuda()`` / ``.to()``).
    """

    SUPPORTS_16BIT = False            # whether forward() has a fp16 / bf16 storage path (set by the model classes that do)
    storage_dtype = torch.float32      # activation / weight storage of the forward: float32, or float16 / bfloat16 (set_storage)

    def __init__(self):
        super().__init__()
        self._hip_packed = None

    def set_storage(self, dtype):
        """Run this network with ``dtype`` activations and weights: torch.float32 (default; convs on the split-fp16 or fp32 matrix
        cores), or torch.float16 / torch.bfloat16 = the 16-bit storage path (BASELINE configs[2] / configs[4]: one MFMA per product, fp32
        accumulation and epilogue).  Parameters stay fp32 ``nn.Parameter``s -- the state_dict is unchanged -- and are rounded once when
        they are packed."""
        if dtype not in (torch.float32, torch.float16, torch.bfloat16):
            raise _lib.ArsegError(f"unsupported storage dtype {dtype}")
        if dtype != torch.float32 and not self.SUPPORTS_16BIT:
            raise _lib.ArsegError(f"{type(self).__name__} has no 16-bit storage path (only the BiSeNet family does); use torch.float32")
        self.storage_dtype = dtype
        return self

    def _apply(self, fn, *args, **kwargs):
        self._hip_packed = None
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._hip_packed = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _pack(self, device):  # pragma: no cover - overridden
        raise NotImplementedError

    def packed(self):
        if self._hip_packed is None:
            p = next(self.parameters())
            if not p.is_cuda:
                raise _lib.ArsegError(f"{type(self).__name__}: parameters are on {p.device}; this implementation runs on the GPU "
                                      "only (call .cuda()); there is no CPU fallback")
            with torch.no_grad():
                self._hip_packed = self._pack(p.device)
        return self._hip_packed

    def _check_inference(self):
        if self.training:
            raise _lib.ArsegError(f"{type(self).__name__} is inference-only (BatchNorm folded, dropout = identity): call .eval() "
                                  "first, as the reference's evaluation.py does")
