"""arseg_amd -- MI355X-native (gfx950) implementation of AR-Seg's LR-branch inference hot path.

The package mirrors the reference's operator interface for this path only
(``model/attention.py``, ``model/pspnet.py``, ``model/bisenet.py``, ``evaluation.warpFeature``),
backed by hand-written HIP kernels behind the C ABI declared in ``include/arseg_hip.h``
(``ar-seg_amd/lib/libarseg_hip.so``).  There is no CPU fallback: every op raises if the
library is missing or a tensor is not on the GPU.
"""
__version__ = "0.1.0"
