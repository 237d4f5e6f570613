"""sparsebench_amd -- MI355X-native hot path of SparseBench's CG solver.

Layout
  csrc/     hand-written HIP kernels (gfx950) + the C-ABI in include/sbhip.h
  host/     C host side mirroring the reference's solver/matrix/comm interface
  capi.py   ctypes binding of libsbhip.so        (plumbing)
  hostapi.py ctypes binding of libsparsebench_host.so (plumbing)

The compute path is the HIP library only; importing this package never touches
oracle/ and there is no CPU fallback.
"""
from . import capi  # noqa: F401

__all__ = ["capi", "hostapi"]
__version__ = "0.1.0"
