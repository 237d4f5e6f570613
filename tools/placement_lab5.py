#!/usr/bin/env python3
"""WHICH memory, not where in it: the same matrix in a series of fresh slabs (earlier ones kept allocated), and the loop's vectors
in a series of fresh allocations (earlier ones kept), stand-alone SpMV and CG step timed for each.  One process.
usage: placement_lab5.py [slabs=10] [vector_sets=8]"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

nslab = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nvec = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
dx, dy = DeviceVector.from_host(np.random.default_rng(1).standard_normal(p.nc)), DeviceVector(p.nr)
ea, eb = L.sb_event_create(), L.sb_event_create()


def spmv_alone(reps=12):
    for _ in range(3):
        L.sb_spmv_native(p.matrix, dx.ptr, dy.ptr)
    best = 1e9
    for _ in range(3):
        L.sb_event_record(ea)
        for _ in range(reps):
            L.sb_spmv_native(p.matrix, dx.ptr, dy.ptr)
        L.sb_event_record(eb)
        best = min(best, 1e3 * L.sb_event_elapsed_ms(ea, eb) / reps)
    return best


def cg_step(cg):
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
    return 1e6 * best


mp = (C.c_uint64 * 4)()
L.sb_matrix_debug_ptrs(p.matrix, mp)
print("matrix where hipMalloc put it (val %x): SpMV alone %.2f us" % (mp[1], spmv_alone()), flush=True)
for i in range(nslab):
    L.sb_matrix_place_fresh(p.matrix)
    L.sb_matrix_debug_ptrs(p.matrix, mp)
    print("matrix in fresh slab %2d (val %x): SpMV alone %.2f us" % (i, mp[1], spmv_alone()), flush=True)
kept = []
for i in range(nvec):
    cg = hostapi.CG(p)
    vp = (C.c_uint64 * 8)()
    L.sb_cg_debug_ptrs(cg.ptr, vp)
    print("vectors in fresh allocation %2d (r %x): %.2f us per CG step (matrix in the last slab)" % (i, vp[0], cg_step(cg)), flush=True)
    kept.append(cg)  # keep it: the next set lands on other memory
