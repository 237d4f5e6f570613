#!/usr/bin/env python3
"""Which allocation decides the speed of the section-8d loop?  One process; for a sweep of dummy device allocations in front of
everything (`front`) or between the matrix and the CG vectors (`mid`), upload, time 2 x 120 clean CG steps and a stand-alone SpMV,
and print the device addresses of the streamed arrays and of the vectors next to the times.
usage: placement_lab2.py [front|mid] [max_mb=1024] [step_mb=32]"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

where = sys.argv[1] if len(sys.argv) > 1 else "front"
mx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
step = int(sys.argv[3]) if len(sys.argv) > 3 else 32
L = capi.init(0)
MB = 1 << 20
ea, eb = L.sb_event_create(), L.sb_event_create()
for d in list(range(0, mx + 1, step)) + [0]:
    d1 = L.sb_malloc(d * MB) if d and where == "front" else None
    p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
    assert p.use_packed(0) == 0
    d2 = L.sb_malloc(d * MB) if d and where == "mid" else None
    cg = hostapi.CG(p)
    best = 1e9
    for seg in range(2):
        cg.start(itermax=128, eps=0.0)
        cg.run_iters(6)
        L.sb_sync()
        t0 = time.perf_counter()
        cg.run_iters(120)
        L.sb_sync()
        best = min(best, (time.perf_counter() - t0) / 120)
        cg.finish()
    mp = (C.c_uint64 * 4)()
    vp = (C.c_uint64 * 8)()
    L.sb_matrix_debug_ptrs(p.matrix, mp)
    L.sb_cg_debug_ptrs(cg.ptr, vp)
    dx, dy = DeviceVector(p.nc), DeviceVector(p.nr)
    for _ in range(3):
        L.sb_spmv_native(p.matrix, dx.ptr, dy.ptr)
    L.sb_event_record(ea)
    for _ in range(12):
        L.sb_spmv_native(p.matrix, dx.ptr, dy.ptr)
    L.sb_event_record(eb)
    alone = 1e3 * L.sb_event_elapsed_ms(ea, eb) / 12
    print("%s %4d MB: %.2f us per CG step, SpMV alone %.2f us | colInd %x val %x chunkPtr %x | r %x p %x Ap %x x %x" % (
        where, d, 1e6 * best, alone, mp[0], mp[1], mp[2], vp[0], vp[1], vp[3], vp[4]), flush=True)
    dx.free(), dy.free(), cg.free(), p.free()
    for q in (d1, d2):
        if q:
            L.sb_free(q)
