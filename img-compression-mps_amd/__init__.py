"""MI355X-native NDMPS hot path (encode -> truncate -> reconstruct).

Drop-in for ``imgcompressionmps.core.ndmps.NDMPS`` of Alandroid/img-compression-mps;
all arithmetic runs in hand-written HIP kernels (libndmps_hip.so, C ABI in
include/ndmps_hip.h).  There is no CPU fallback: importing works without a GPU (so the
ABI can be inspected), every compute call raises if the library or the device is missing.
"""
from .core.ndmps import NDMPS  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["NDMPS"]
