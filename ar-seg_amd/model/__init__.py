"""Mirrors of the reference's model package for the LR-branch inference hot path."""
from .attention import MyAttention, f_similar, f_weighting  # noqa: F401
from .pspnet import PSPNet, PSPNetWithFuse  # noqa: F401
from .bisenet import BiSeNetV1, BiSeNetV1WithFuse  # noqa: F401
from . import pspnet_semseg  # noqa: F401  (Cityscapes PSPNet-18: pspnet_semseg.PSPNet / pspnet_semseg.PSPNetWithFuse)
