"""Drop-in for the third-party ``localAttention`` extension the reference imports
(model/attention.py:7-11): same five function names and Tensor-in / Tensor-out signatures.

To use it from unmodified reference code put this package's directory on ``sys.path`` ahead of any
CUDA build (``import localAttention`` then resolves here), see INTEGRATION.md.  The forward pair runs
the HIP kernels of libarseg_hip.so; the backward trio belongs to training, which is outside the
LR-branch inference hot path, and raises.
"""
from arseg_amd import ops as _ops


def similar_forward(x_ori, x_loc, kH, kW):
    return _ops.local_similar(x_ori, x_loc, int(kH), int(kW))


def weighting_forward(x_ori, x_weight, kH, kW):
    return _ops.local_weighting(x_ori, x_weight, int(kH), int(kW))


def _training_only(*args, **kwargs):
    raise NotImplementedError("localAttention backward is training-only and not part of the inference hot path")


similar_backward = weighting_backward_ori = weighting_backward_weight = _training_only
