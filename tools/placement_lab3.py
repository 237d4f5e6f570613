#!/usr/bin/env python3
"""Does the in-CG speed of the section-8d SpMV follow WHERE THE GATHERED VECTOR lives?  (placement_lab2: the stand-alone SpMV is
the same at every allocation pattern of a process while the CG step moves in steps of ~7.7 us.)  One process, one matrix; the x
vector is taken from a 2 GiB slab at offsets k * step; before every timed SpMV it is re-written by another kernel (as the p update
does inside CG: the gather then finds it in no L2), y and the source vector stay put.  Prints SpMV time by offset.
usage: placement_lab3.py [step_mb=8] [count=96]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

step = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 96
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
n = p.nc
src = DeviceVector.from_host(np.random.default_rng(1).standard_normal(n))
dy = DeviceVector(p.nr)
slab_bytes = int(count * step * (1 << 20)) + n * 8 + (4 << 20)
slab = L.sb_malloc(slab_bytes)
ea, eb = L.sb_event_create(), L.sb_event_create()


def timed(xptr, reps=8):
    tot = 0.0
    for i in range(reps + 2):
        L.sb_waxpby(n, 1.0, src.ptr, 0.0, src.ptr, xptr)  # x rewritten by another kernel, as the p update does
        L.sb_event_record(ea)
        L.sb_spmv_native(p.matrix, xptr, dy.ptr)
        L.sb_event_record(eb)
        ms = L.sb_event_elapsed_ms(ea, eb)
        if i >= 2:
            tot += ms
    return 1e3 * tot / reps


own = DeviceVector(n)
print("x in a buffer of its own (hipMalloc): %.2f us" % timed(own.ptr), flush=True)
vals = []
for k in range(count):
    off = int(k * step * (1 << 20)) & ~255
    t = timed(slab + off)
    vals.append(t)
    print("x at slab + %7.1f MB (address %x): %.2f us" % (off / 2 ** 20, slab + off, t), flush=True)
v = np.array(vals)
print("min %.2f max %.2f median %.2f; levels (rounded to 1 us): %s" % (v.min(), v.max(), np.median(v), sorted(set(np.round(v).astype(int).tolist()))))
