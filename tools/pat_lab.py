#!/usr/bin/env python3
"""pat_lab.py [n] [sigma] -- stand-alone timing of every SpMV kernel mode on the HPCG matrix
(back-to-back launches, HIP events).  SBHIP_LIBRARY=<path> times a lab build of libsbhip.so
(e.g. one compiled with -DSB_LAB=1 to knock out a phase of a kernel)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402  (matrix generation only)
from sparsebench_amd import capi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 256
modes = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [3, 2, 1, 0]
L = capi.init(0)
g = po.GMatrix.generate(n, n, n)
s = g.to_scs(64, sigma)
arrs = [np.ascontiguousarray(a) for a in (s.chunkPtr, s.chunkLens, s.colInd, s.val, s.oldToNewPerm, s.newToOldPerm)]
m = L.sb_scs_upload(s.nr, s.nc, 64, sigma, s.nChunks, s.nElems, *[a.ctypes.data_as(C.c_void_p) for a in arrs])
x, y = DeviceVector.from_host(np.ones(s.nc)), DeviceVector(s.nr)
for mode in modes:
    L.sb_matrix_use_packed(m, mode)
    if L.sb_matrix_packed_mode(m) != mode:
        continue
    for _ in range(5):
        L.sb_spmv_native(m, x.ptr, y.ptr)
    a, b = L.sb_event_create(), L.sb_event_create()
    best = 1e9
    for rep in range(3):
        L.sb_event_record(a)
        for _ in range(100):
            L.sb_spmv_native(m, x.ptr, y.ptr)
        L.sb_event_record(b)
        best = min(best, 1e3 * L.sb_event_elapsed_ms(a, b) / 100)
    print("%s mode %d: %.2f us  (moves %.1f MB)" % (os.environ.get("SBHIP_LIBRARY", "product"), mode, best,
                                                   L.sb_matrix_stream_bytes(m) / 1e6))
    if hasattr(L, "sb_spmv_native_dot"):  # the same launch with the fused level-0 partials of x . y (what CG runs)
        q = DeviceVector.from_host(np.zeros(4 * ((s.nr + 255) // 256)))
        if L.sb_spmv_native_dot(m, x.ptr, y.ptr, q.ptr):
            best = 1e9
            for rep in range(3):
                L.sb_event_record(a)
                for _ in range(100):
                    L.sb_spmv_native_dot(m, x.ptr, y.ptr, q.ptr)
                L.sb_event_record(b)
                best = min(best, 1e3 * L.sb_event_elapsed_ms(a, b) / 100)
            print("   ... with the fused dot: %.2f us" % best)
        q.free()

# lab build with per-tile timestamps (sb_lab_prof): phases of the pattern kernel (the first of `modes`), in us
if hasattr(L, "sb_lab_prof"):
    L.sb_matrix_use_packed(m, modes[0])
    L.sb_spmv_native(m, x.ptr, y.ptr)
    L.sb_spmv_native(m, x.ptr, y.ptr)
    prof = np.zeros(8192 * 8, dtype=np.int64)
    L.sb_lab_prof(prof.ctypes.data_as(C.c_void_p))
    cpt = 8 if s.nChunks > 4 else 4
    nT = min(8192, (s.nChunks + cpt - 1) // cpt)
    p = prof.reshape(8192, 8)[:nT, :6].astype(np.float64) / 100.0  # wall_clock64: 100 MHz
    t0 = p[:, 0].min()
    print("mode %d: tiles %d; kernel span %.2f us" % (L.sb_matrix_packed_mode(m), nT, p[:, 5].max() - t0))
    names = ["start->header", "header->loads back", "loads->barrier passed", "barrier->chunk 0 summed", "chunk 0->chunk 1 summed"]
    for i, nm in enumerate(names):
        d = p[:, i + 1] - p[:, i]
        print("  %-26s mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f us" % (nm, d.mean(), *np.percentile(d, [10, 50, 90])))
    life = p[:, 5] - p[:, 0]
    print("  tile life                  mean %.2f  p50 %.2f  p90 %.2f us" % (life.mean(), *np.percentile(life, [50, 90])))
    starts = np.sort(p[:, 0] - t0)
    print("  start times: p25 %.2f p50 %.2f p75 %.2f p100 %.2f us" % tuple(np.percentile(starts, [25, 50, 75, 100])))
