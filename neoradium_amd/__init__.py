"""neoradium_amd -- MI355X-native PDSCH link-level hot path behind the NeoRadium class surface.

The compute path is libnrx.so (hand-written HIP for gfx950, C ABI in include/nrx.h).  There is no CPU fallback:
calling any operator without the built library or without a GPU raises.
"""
__version__ = '0.1.0'
