"""``import localAttention`` for unmodified reference code (model/attention.py:7-11).

This directory holds nothing else: put IT (not the package directory) on ``sys.path`` ahead of any CUDA build of the
third-party extension and the reference's ``from localAttention import similar_forward, ...`` resolves here, while its
own ``model`` / ``dataset`` packages keep resolving to the reference.  The functions are those of
``arseg_amd.localAttention`` (forward pair on libarseg_hip.so; the backward trio is training-only and raises).
"""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))      # the repository root holds the `arseg_amd` import shim
if _root not in _sys.path:
    _sys.path.append(_root)

from arseg_amd.localAttention import (similar_backward, similar_forward, weighting_backward_ori,  # noqa: E402,F401
                                      weighting_backward_weight, weighting_forward)
