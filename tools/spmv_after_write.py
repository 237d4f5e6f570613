#!/usr/bin/env python3
"""spmv_after_write.py [n] [sigma] -- the default SpMV timed (HIP events around each launch) when x was just rewritten by
another kernel (as p is inside CG) against back-to-back launches on an untouched x."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = capi.init(0)
p = hostapi.Problem("generate", n, n, n, fmt="scs", Cc=64, sigma=sigma)
x, y, z = DeviceVector.from_host(np.ones(p.nc)), DeviceVector(p.nr), DeviceVector.from_host(np.ones(p.nc))
reps = 100
evs = [(L.sb_event_create(), L.sb_event_create()) for _ in range(reps)]
for label, dirty in (("x untouched", 0), ("x rewritten before every launch", 1), ("another 50 MB streamed before every launch", 2)):
    for _ in range(5):
        L.sb_spmv_native(p.matrix, x.ptr, y.ptr)
    for a, b in evs:
        if dirty == 1:
            L.sb_waxpby(p.nr, 1.0, x.ptr, 0.0, x.ptr, x.ptr)
        elif dirty == 2:
            L.sb_waxpby(p.nr, 1.0, z.ptr, 0.0, z.ptr, z.ptr)
        L.sb_event_record(a)
        L.sb_spmv_native(p.matrix, x.ptr, y.ptr)
        L.sb_event_record(b)
    L.sb_sync()
    t = sorted(1e3 * L.sb_event_elapsed_ms(a, b) for a, b in evs)
    print("mode %d  %-45s median %.2f us  min %.2f" % (p.pack_info()["mode"], label, t[len(t) // 2], t[0]), flush=True)
