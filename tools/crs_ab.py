#!/usr/bin/env python3
"""native CRS SpMV A/B: spmv_crs_split (equal nonzero windows, default) vs spmv_crs_stream (row blocks, SB_CRS_KERNEL=stream)
on HPCG n^3 and on the irregular stand-in, stand-alone, bit-checked against the oracle's CRS loop.
usage: crs_ab.py [hpcg_n=128] [irregular_n=80] [reps=50]   (run once per SB_CRS_KERNEL setting: the choice is made at upload)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

hn = int(sys.argv[1]) if len(sys.argv) > 1 else 128
irn = int(sys.argv[2]) if len(sys.argv) > 2 else 80
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
L = capi.init(0)
which = os.environ.get("SB_CRS_KERNEL", "split")
for name, prob in (("hpcg %d^3" % hn, hostapi.Problem("generate", hn, hn, hn, fmt="crs")),
                   ("irregular %d^3 nodes" % irn, hostapi.Problem("irregular", irn, irn, irn, fmt="crs"))):
    m = prob.matrix
    assert prob.use_packed(0) == 0
    rng = np.random.default_rng(5)
    x = rng.standard_normal(prob.nc)

    dx, dy = DeviceVector.from_host(x), DeviceVector(prob.nr)
    for _ in range(5):
        L.sb_spmv_native(m, dx.ptr, dy.ptr)
    best = 1e30
    for _ in range(5):
        a, b = L.sb_event_create(), L.sb_event_create()
        L.sb_event_record(a)
        for _ in range(reps):
            L.sb_spmv_native(m, dx.ptr, dy.ptr)
        L.sb_event_record(b)
        best = min(best, 1e3 * L.sb_event_elapsed_ms(a, b) / reps)
    y = dy.get()
    # the CPU's loop, row by row left to right (src/matrix-CRS.c:54-64), vectorised over rows of equal length
    rp, ci, va = prob.array("rowPtr").astype(np.int64), prob.array("crs_colInd"), prob.values()
    ref = np.zeros(prob.nr)
    lens = rp[1:] - rp[:-1]
    for j in range(int(lens.max())):
        rows = np.nonzero(lens > j)[0]
        k = rp[rows] + j
        ref[rows] = ref[rows] + va[k] * x[ci[k]]
    alg = L.sb_matrix_spmv_bytes(m)
    print("%-8s %-24s %8.1f us  %6.0f GB/s  frac %.3f  bit-exact=%s" % (which, name, best, alg / best / 1e3, alg / best / 1e3 / 8000,
                                                                       bool(np.array_equal(y, ref))), flush=True)
    dx.free(), dy.free()
    prob.free()
