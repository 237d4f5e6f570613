#!/usr/bin/env python3
"""pat_report.py [n] [sigma] -- upload the HPCG matrix with SB_PACK_REPORT=1 and print what the pack levels built,
then compare every kernel mode's y with the oracle (bitwise)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("SB_PACK_REPORT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from sparsebench_amd import capi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dims = (n, n, n) if len(sys.argv) <= 3 else tuple(int(v) for v in sys.argv[3].split("x"))
L = capi.init(0)
g = po.GMatrix.generate(*dims)
s = g.to_scs(64, sigma)
arrs = [np.ascontiguousarray(a) for a in (s.chunkPtr, s.chunkLens, s.colInd, s.val, s.oldToNewPerm, s.newToOldPerm)]
m = L.sb_scs_upload(s.nr, s.nc, 64, sigma, s.nChunks, s.nElems, *[a.ctypes.data_as(C.c_void_p) for a in arrs])
mch = C.c_uint32(0)
print("default mode", L.sb_matrix_packed_mode(m), "programs", L.sb_matrix_row_programs(m, C.byref(mch)), "masked chunks",
      mch.value, "of", s.nChunks)
rng = np.random.default_rng(5)
xh = rng.standard_normal(s.nc)
exp = g.spmv(xh)
x, y = DeviceVector.from_host(xh), DeviceVector(s.nr)
for mode in (5, 3, 2, 1, 0):
    L.sb_matrix_use_packed(m, mode)
    if L.sb_matrix_packed_mode(m) != mode:
        continue
    y.set(np.full(s.nr, 7.0))
    L.sb_spmv(m, x.ptr, y.ptr)
    got = y.get()
    bad = np.flatnonzero(got.view(np.uint64) != exp.view(np.uint64))
    print("mode", mode, "moves %.1f MB" % (L.sb_matrix_stream_bytes(m) / 1e6), "mismatching rows:", len(bad), bad[:8])
