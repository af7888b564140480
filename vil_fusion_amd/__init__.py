"""vil_fusion_amd — MI355X-native sliding-window back-end for VIL_Fusion (hot path only).

Product path: libvilfusion_hip.so (hand-written HIP for gfx950) behind the C ABI of include/vilfusion.h.
There is NO CPU fallback: importing the solver classes without the built HIP library raises.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
