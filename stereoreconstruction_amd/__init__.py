"""MI355X-native dense matching-cost / support-weight / WTA path of StereoReconstruction.

The compute path is libstereo_recon_hip.so (hand-written HIP kernels for gfx950
behind the C-ABI of include/stereo_recon_hip.h).  ``capi`` is its ctypes binding;
``synthetic`` generates the deterministic benchmark inputs.  Nothing here falls
back to a CPU implementation.
"""
from . import capi, synthetic  # noqa: F401

__all__ = ["capi", "synthetic"]
