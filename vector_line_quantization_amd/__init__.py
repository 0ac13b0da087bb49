"""MI355X-native IVF(PQ) list-scan search path.

The product is the HIP shared library `csrc/libvlq_ivfpq.so` behind the C ABI of
include/vlq_ivfpq.h; this package is the thin ctypes loader used by the tests,
bench.py and the multi-GPU driver.  There is no CPU implementation in here: if
the library is missing or no HIP device is present, calls fail loudly.
"""
from ._lib import VlqError, build_library, device_count, lib, library_path  # noqa: F401
from .index import GpuIVFPQ, GpuVLQ  # noqa: F401

__all__ = ["GpuIVFPQ", "GpuVLQ", "VlqError", "build_library", "device_count", "lib", "library_path"]
