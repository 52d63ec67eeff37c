"""ciaosr_amd -- MI355X-native implementation of CiaoSR's LocalImplicitSR forward path.

Python host code mirrors the reference's restorer/backbone plugin surface; the arithmetic of the
head runs in libciaosr_hip.so (hand-written gfx950 HIP kernels behind a C ABI, include/ciaosr_hip.h).
"""
from .registry import register, build_model, build_backbone, build_component, build_loss  # noqa: F401
from .encoders import RDN, EDSR, SwinIR

register('RDN')(RDN)
register('EDSR')(EDSR)
register('SwinIR')(SwinIR)

from .mlp import MLPRefiner  # noqa: E402,F401
from .nonlocal_attn import CrossScaleAttention  # noqa: E402,F401
from .implicit_net import (LocalImplicitSRNet, LocalImplicitSRRDN, LocalImplicitSREDSR,  # noqa: E402,F401
                           LocalImplicitSRSWINIR)
from .restorer import BasicRestorer, CiaoSR  # noqa: E402,F401

__version__ = '0.1.0'
