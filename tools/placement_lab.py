#!/usr/bin/env python3
"""How fast does the section-8d kernel stream the SAME matrix from different places in device memory?  One process, one upload;
the reference-layout arrays are moved through a grid of offsets inside their slab (sb_matrix_place) and `reps` stand-alone
launches are timed at every point (HIP events on the layer's stream).  Prints the map, the best and the worst point, and the
time at hipMalloc's own placement.  usage: placement_lab.py [fmt=scs] [step_mb=16] [reps=12] [n=128]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

fmt = sys.argv[1] if len(sys.argv) > 1 else "scs"
step = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n = int(sys.argv[4]) if len(sys.argv) > 4 else 128
L = capi.init(0)
p = hostapi.Problem("generate", n, n, n, fmt=fmt, Cc=64, sigma=256)
assert p.use_packed(0) == 0
x = np.random.default_rng(1).standard_normal(p.nc)
dx, dy = DeviceVector.from_host(x), DeviceVector(p.nr)
ea, eb = L.sb_event_create(), L.sb_event_create()
alg = p.spmv_bytes()


def timed():
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


base = timed()
y0 = dy.get().copy()
print("placement_lab %s %d^3: hipMalloc's own placement: %.2f us per launch (%.3f of 8 TB/s)" % (fmt, n, base, alg / base / 8e6), flush=True)
offs = list(range(0, 257, step))
grid = np.zeros((len(offs), len(offs)))
print("rows: colInd offset [MB]; columns: val offset [MB] " + " ".join("%6d" % o for o in offs))
for i, co in enumerate(offs):
    for j, vo in enumerate(offs):
        L.sb_matrix_place(p.matrix, co, vo)
        grid[i, j] = timed()
    print("%4d MB: " % co + " ".join("%6.1f" % v for v in grid[i]), flush=True)
assert np.array_equal(dy.get(), y0)  # (same product from every place)
i, j = np.unravel_index(np.argmin(grid), grid.shape)
k, l = np.unravel_index(np.argmax(grid), grid.shape)
print("best  %.2f us (%.3f of 8 TB/s) at colInd +%d MB, val +%d MB" % (grid[i, j], alg / grid[i, j] / 8e6, offs[i], offs[j]))
print("worst %.2f us (%.3f) at colInd +%d MB, val +%d MB; median %.2f us" % (grid[k, l], alg / grid[k, l] / 8e6, offs[k], offs[l], float(np.median(grid))))
print("by val offset (min over colInd offsets): " + " ".join("%d:%.1f" % (o, grid[:, j].min()) for j, o in enumerate(offs)))
print("by colInd offset (min over val offsets): " + " ".join("%d:%.1f" % (o, grid[i].min()) for i, o in enumerate(offs)))
