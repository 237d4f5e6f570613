#!/usr/bin/env python3
"""mode_times.py <file|generate|irregular> nx ny nz fmt C sigma -- time y = A x (sb_spmv_native, back to back) in every
kernel mode the matrix has"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

name, nx, ny, nz, fmt, Cc, sigma = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
L = capi.init(0)
p = hostapi.Problem(name, nx, ny, nz, fmt=fmt, Cc=Cc, sigma=sigma)
print("rows", p.nr, "nnz", p.nnzTrue, "default mode", p.pack_info())
x, y = DeviceVector.from_host(np.ones(p.nc)), DeviceVector(p.nr)
for mode in (5, 3, 2, 1, 0):
    if p.use_packed(mode) != mode:
        continue
    for _ in range(3):
        L.sb_spmv_native(p.matrix, x.ptr, y.ptr)
    a, b = L.sb_event_create(), L.sb_event_create()
    L.sb_event_record(a)
    for _ in range(50):
        L.sb_spmv_native(p.matrix, x.ptr, y.ptr)
    L.sb_event_record(b)
    print("mode %d: %.2f us (moves %.3f MB)" % (mode, 1e3 * L.sb_event_elapsed_ms(a, b) / 50, p.stream_bytes() / 1e6), flush=True)
