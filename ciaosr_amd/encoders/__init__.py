from .rdn import RDN
from .edsr import EDSR
from .swinir import SwinIR

__all__ = ['RDN', 'EDSR', 'SwinIR']
