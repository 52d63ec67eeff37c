from .rdn import RDN
from .edsr import EDSR

__all__ = ['RDN', 'EDSR']
