from ciaosr_amd.implicit_net import (LocalImplicitSRNet, LocalImplicitSRRDN, LocalImplicitSREDSR,  # noqa: F401
                                     LocalImplicitSRSWINIR)
