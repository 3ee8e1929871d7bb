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

    def __init__(self):
        super().__init__()
        self._hip_packed = None

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
