#!/usr/bin/env python3
"""configs[4] (irregular-nnz stress; Flan_1565 stand-in = the product's "irregular" generator,
host/sbh_irregular.c): CRS vs Sell-C-sigma SpMV on one GPU, every available kernel mode,
checked bit for bit against the oracle's CRS loop.  usage: irregular_bench.py [n=80] [reps=30]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sigmas = [int(s) for s in sys.argv[3].split(",") if s.strip().isdigit()] if len(sys.argv) > 3 else [1, 256, 4096]
L = capi.init(0)
rows = []


def run(prob, x, yref, label, mode=None):
    m = prob.matrix
    if mode is not None:
        if prob.use_packed(mode) != mode:
            return
    perm = L.sb_matrix_is_permuted(m)
    dx, dy = DeviceVector.from_host(x), DeviceVector(prob.nr)
    if perm:  # time the native (permuted-order) kernel, as CG runs it
        dxp = DeviceVector(prob.nc)
        L.sb_permute(m, dx.ptr, dxp.ptr)
    else:
        dxp = dx
    for _ in range(3):
        L.sb_spmv_native(m, dxp.ptr, dy.ptr)
    a, b = L.sb_event_create(), L.sb_event_create()
    L.sb_event_record(a)
    for _ in range(reps):
        L.sb_spmv_native(m, dxp.ptr, dy.ptr)
    L.sb_event_record(b)
    us = 1e3 * L.sb_event_elapsed_ms(a, b) / reps
    L.sb_spmv(m, dx.ptr, dy.ptr)
    ok = bool(np.array_equal(dy.get(), yref))
    alg, moved = L.sb_matrix_spmv_bytes(m), L.sb_matrix_stream_bytes(m)
    r = {"kernel": label, "mode": L.sb_matrix_packed_mode(m), "us": us, "alg_MB": alg / 1e6, "moved_MB": moved / 1e6,
         "alg_GBs": alg / us / 1e3, "frac_alg": alg / us / 1e3 / 8000, "moved_GBs": moved / us / 1e3, "bit_exact": ok}
    rows.append(r)
    print("%-28s mode %d %8.1f us  alg %7.1f MB %6.0f GB/s (%.3f)  moved %7.1f MB %6.0f GB/s  bit-exact=%s" % (
        label, r["mode"], us, r["alg_MB"], r["alg_GBs"], r["frac_alg"], r["moved_MB"], r["moved_GBs"], ok), flush=True)
    dx.free(), dy.free()
    if perm:
        dxp.free()


t0 = time.time()
crs = hostapi.Problem("irregular", n, n, n, fmt="crs")
print("irregular %d^3 nodes: %d rows, %d nnz (%.1f per row), setup %.1fs" % (n, crs.nr, crs.nnzTrue, crs.nnzTrue / crs.nr,
                                                                               time.time() - t0), flush=True)
rp = crs.array("rowPtr").copy()
col, val = crs.gm_entries()
g = po.GMatrix.from_csr(rp, col, val, nc=crs.nc)
rng = np.random.default_rng(7)
x = rng.standard_normal(crs.nc)
yref = g.spmv(x)
for mode in (0, 3):
    run(crs, x, yref, "CRS", mode)
crs.free()
for sigma in sigmas:
    t0 = time.time()
    scs = hostapi.Problem("irregular", n, n, n, fmt="scs", Cc=64, sigma=sigma)
    fill = scs.nnzTrue / scs.nElems
    print("SCS C=64 sigma=%d: fill %.3f, nElems %d, setup %.1fs, pack %r" % (sigma, fill, scs.nElems, time.time() - t0, scs.pack_info()), flush=True)
    for mode in (0, 1, 2, 3):
        run(scs, x, yref, "SCS C=64 sigma=%d" % sigma, mode)
    scs.free()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "irregular_bench_%d.json" % n)
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rows, open(out, "w"), indent=1)
