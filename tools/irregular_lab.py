#!/usr/bin/env python3
"""configs[4] stand-in (SuiteSparse Flan_1565 is not available offline): a synthetic
irregular SPD-patterned matrix -- 3x3-block FE-like rows whose lengths vary 4x, columns
clustered near the diagonal plus a few far couplings -- CRS vs Sell-C-sigma on one GPU.
Inputs to the product come from numpy via the oracle's converter (test infrastructure);
the kernels timed are the product's."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from sparsebench_amd import capi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

vp = C.c_void_p
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000
rng = np.random.default_rng(5)
nodes = nr // 3
deg = rng.integers(8, 33, size=nodes)            # neighbours per node: 8..32 -> 24..96 nnz/row
lens = np.repeat(deg * 3, 3)
rp = np.zeros(nr + 1, dtype=np.uint32)
rp[1:] = np.cumsum(lens)
nnz = int(rp[-1])
cols = np.empty(nnz, dtype=np.uint32)
pos = 0
for v in range(nodes):                            # vectorised per node: neighbours near v, 5 % far
    k = deg[v]
    nb = v + rng.integers(-400, 401, size=k)
    far = rng.random(k) < 0.05
    nb[far] = rng.integers(0, nodes, size=int(far.sum()))
    nb = np.unique(np.clip(nb, 0, nodes - 1))
    while len(nb) < k:
        nb = np.unique(np.concatenate([nb, rng.integers(0, nodes, size=k - len(nb))]))
    c3 = (nb[:, None] * 3 + np.arange(3)[None, :]).ravel().astype(np.uint32)
    for d in range(3):
        cols[pos:pos + 3 * k] = c3
        pos += 3 * k
vals = rng.standard_normal(nnz)
print("rows %d nnz %d (%.1f per row, min %d max %d)" % (nr, nnz, nnz / nr, lens.min(), lens.max()), flush=True)
g = po.GMatrix.from_csr(rp, cols, vals, nc=nr)
L = capi.init(0)
x = rng.standard_normal(nr)
yref = g.spmv(x)


def run(m, reps=50):
    dx, dy = DeviceVector.from_host(x), DeviceVector(nr)
    for _ in range(3):
        L.sb_spmv_native(m, dx.ptr, dy.ptr)
    a, b = L.sb_event_create(), L.sb_event_create()
    L.sb_event_record(a)
    for _ in range(reps):
        L.sb_spmv_native(m, dx.ptr, dy.ptr)
    L.sb_event_record(b)
    us = 1e3 * L.sb_event_elapsed_ms(a, b) / reps
    L.sb_spmv(m, dx.ptr, dy.ptr)
    ok = np.array_equal(dy.get(), yref)
    dx.free(), dy.free()
    return us, ok


def p(a):
    return np.ascontiguousarray(a).ctypes.data_as(vp)


m = L.sb_crs_upload(nr, nr, p(g.rowPtr), p(g.col), p(g.val))
us, ok = run(m)
B = L.sb_matrix_spmv_bytes(m)
print("CRS                      %8.1f us  %7.0f GB/s  bit-exact=%s" % (us, B / us / 1e3, ok))
L.sb_matrix_free(m)
for sigma in (1, 64, 1024, 65536):
    s = g.to_scs(64, sigma)
    arrs = [np.ascontiguousarray(a) for a in (s.chunkPtr, s.chunkLens, s.colInd, s.val, s.oldToNewPerm, s.newToOldPerm)]
    m = L.sb_scs_upload(s.nr, s.nc, 64, sigma, s.nChunks, s.nElems, *[a.ctypes.data_as(vp) for a in arrs])
    beta = nnz / s.nElems
    for mode in (0, 1, 2):
        L.sb_matrix_use_packed(m, mode)
        if L.sb_matrix_packed_mode(m) != mode:
            continue
        us, ok = run(m)
        print("SCS C=64 sigma=%-6d mode %d %8.1f us  %7.0f GB/s(algorithmic, CRS bytes)  fill=%.3f moved=%.0f MB bit-exact=%s" % (
            sigma, mode, us, B / us / 1e3, beta, L.sb_matrix_stream_bytes(m) / 1e6, ok))
    L.sb_matrix_free(m)
