#!/usr/bin/env python3
"""profiling driver for configs[4]: `reps` launches of ONE format's SpMV on the irregular stand-in.
usage: irregular_prof_run.py <crs|scs> <sigma> [n=80] [reps=20] [mode=-1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

fmt, sigma = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
mode = int(sys.argv[5]) if len(sys.argv) > 5 else -1
L = capi.init(0)
p = hostapi.Problem("irregular", n, n, n, fmt=fmt, Cc=64, sigma=sigma)
if mode >= 0:
    p.use_packed(mode)
x = np.random.default_rng(7).standard_normal(p.nc)
dx, dy = DeviceVector.from_host(x), DeviceVector(p.nr)
for _ in range(reps):
    L.sb_spmv_native(p.matrix, dx.ptr, dy.ptr)
L.sb_sync()
print("done", fmt, sigma, p.nr, p.nnzTrue, L.sb_matrix_packed_mode(p.matrix), L.sb_matrix_stream_bytes(p.matrix))
