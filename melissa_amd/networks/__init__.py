from .dgn_r import DGNRNetwork
from .hl_dgn import HLDGNNetwork
from .l_dgn import LDGNNetwork

__all__ = ["LDGNNetwork", "HLDGNNetwork", "DGNRNetwork"]
