#!/usr/bin/env python3
"""Where does the SCS C=64 SpMV kernel's time go?  Standalone timings (HIP events,
back-to-back launches) of the real HPCG matrix against two synthetic column patterns
of the SAME shape: all columns 0 (no gather traffic) and col = own row (perfectly
coalesced gather).  Also the device's raw streaming-read rate."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = capi.init(0)
print("stream read GB/s: 1GiB %.0f, 700MB %.0f" % (L.sb_debug_stream_read_gbs(1 << 30, 20),
                                                   L.sb_debug_stream_read_gbs(700 << 20, 20)))
p = hostapi.Problem("generate", n, n, n, fmt="scs", Cc=64, sigma=1)
vp = C.c_void_p


def time_spmv(m, nc, nr, reps=50):
    x, y = DeviceVector.from_host(np.ones(nc)), DeviceVector(nr)
    for _ in range(3):
        L.sb_spmv_native(m, x.ptr, y.ptr)
    a, b = L.sb_event_create(), L.sb_event_create()
    L.sb_event_record(a)
    for _ in range(reps):
        L.sb_spmv_native(m, x.ptr, y.ptr)
    L.sb_event_record(b)
    us = 1e3 * L.sb_event_elapsed_ms(a, b) / reps
    x.free(), y.free()
    return us


B = p.spmv_bytes()
us = time_spmv(p.matrix, p.nc, p.nr)
print("hpcg %d^3 real matrix     : %.1f us  %.0f GB/s" % (n, us, B / us / 1e3))
cp, cl = p.array("chunkPtr").copy(), p.array("chunkLens").copy()
val = p.values().copy()
ident = np.arange(p.nr, dtype=np.uint32)
for name, col in (("all columns = 0", np.zeros(p.nElems, dtype=np.uint32)),
                  ("col = own row", None)):
    if col is None:
        lane = np.arange(p.nElems, dtype=np.int64)
        chunk = np.searchsorted(cp, lane, side="right") - 1
        col = np.minimum(chunk * 64 + (lane - cp[chunk]) % 64, p.nr - 1).astype(np.uint32)
    m = L.sb_scs_upload(p.nr, p.nc, 64, 1, p.nChunks, p.nElems, cp.ctypes.data_as(vp),
                        cl.ctypes.data_as(vp), col.ctypes.data_as(vp), val.ctypes.data_as(vp),
                        ident.ctypes.data_as(vp), ident.ctypes.data_as(vp))
    us = time_spmv(m, p.nc, p.nr)
    print("%-26s: %.1f us  %.0f GB/s" % (name, us, B / us / 1e3))
    L.sb_matrix_free(m)
