"""configs[4] on the GPU: irregular-nnz stress, CRS vs Sell-C-sigma (SuiteSparse Flan_1565 is not available
offline; the matrix is the committed stand-in "irregular", host/sbh_irregular.c).  Small size: every kernel mode
bit-equal to the oracle (SpMV and 30+ CG iterations).  Full size (80^3 nodes, 1.5 M rows, 94 M nonzeros): SpMV
bit-equal to the oracle's CRS loop, and the size-independent properties CRS history == Sell-C-1 history,
<y, A x> == <x, A y> to rounding, fused == unfused."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from sparsebench_amd import capi, hostapi
from sparsebench_amd.capi import DeviceVector

pytestmark = pytest.mark.gpu
FORMATS = [("crs", 1), ("scs", 1), ("scs", 4096), ("scs", 65536)]


def oracle_matrix(n):
    p = hostapi.Problem("irregular", n, n, n, fmt="crs", upload=False)
    col, val = p.gm_entries()
    g = po.GMatrix.from_csr(p.array("rowPtr").copy(), col, val, nc=p.nc)
    p.free()
    return g


def gpu_spmv(L, prob, x):
    dx, dy = DeviceVector.from_host(x), DeviceVector(prob.nr)
    L.sb_spmv(prob.matrix, dx.ptr, dy.ptr)
    y = dy.get()
    dx.free(), dy.free()
    return y


@pytest.mark.parametrize("fmt,sigma", FORMATS)
def test_small_spmv_and_cg_bit_equal_to_oracle(gpu, fmt, sigma):
    n = 24  # 41 472 rows, 2.5 M nonzeros
    g = oracle_matrix(n)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(g.nc)
    yref = g.spmv(x)
    o = po.cg(g, itermax=40, fmt=fmt, Cc=64, sigma=sigma, dot="tree", want_x=True)
    # the REFERENCE's own history on this matrix (tests/golden/cg_hist_irregular_ref.json, captured from the reference's
    # reader + solveCG on the stand-in exported as .mtx): north_star's 1e-12 holds against the reference itself here
    import json
    import os
    e = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cg_hist_irregular_ref.json")))["irregular%d" % n]
    ref_rr, ref_pap = np.array([float(v) for v in e["rr"]]), np.array([float(v) for v in e["pAp"]])
    prob = hostapi.Problem("irregular", n, n, n, fmt=fmt, Cc=64, sigma=sigma)
    default = prob.pack_info()["mode"]
    modes = sorted({default, prob.use_packed(0), prob.use_packed(1)})
    for mode in modes:
        assert prob.use_packed(mode) == mode
        assert np.array_equal(gpu_spmv(gpu, prob, x), yref), (fmt, sigma, mode)
        for fused in (True, False):
            cg = hostapi.CG(prob, fused=fused)
            k = cg.solve(40, 0.0)
            rr, pap = cg.history()
            assert k == o["k"] == 40
            assert np.array_equal(rr, o["rr"]) and np.array_equal(pap, o["pAp"]), (fmt, sigma, mode, fused)
            assert np.array_equal(cg.solution(), o["x"][0])
            assert (np.abs(rr - ref_rr) / ref_rr).max() <= 1e-12 and (np.abs(pap - ref_pap) / ref_pap).max() <= 1e-12
            cg.free()
    prob.free()


def test_special_values_propagate_like_the_cpu_loops(gpu):
    """Inf / NaN in x reach exactly the rows the CPU loops let them reach (CRS: rows owning that column;
    Sell-C-sigma additionally through its padding when the column is 0, src/matrix-SCS.c:151-155)"""
    n = 12
    g = oracle_matrix(n)
    x = np.ones(g.nc)
    x[777] = np.inf
    x[0] = np.nan
    for fmt, sigma in (("crs", 1), ("scs", 1), ("scs", 512)):
        ref = g.spmv(x) if fmt == "crs" else g.to_scs(64, sigma).spmv(x)
        prob = hostapi.Problem("irregular", n, n, n, fmt=fmt, Cc=64, sigma=sigma)
        y = gpu_spmv(gpu, prob, x)
        assert np.array_equal(np.isnan(y), np.isnan(ref)) and np.array_equal(y[~np.isnan(y)], ref[~np.isnan(ref)])
        prob.free()


def test_full_size_properties(gpu):
    """BASELINE-scale stand-in: 80^3 nodes = 1 536 000 rows, 94 M nonzeros"""
    n = 80
    g = oracle_matrix(n)
    assert g.nr == 1536000 and 90e6 < g.nnzTrue < 100e6
    rng = np.random.default_rng(3)
    x, z = rng.standard_normal(g.nc), rng.standard_normal(g.nc)
    yref = g.spmv(x)
    hist = {}
    for fmt, sigma in (("crs", 1), ("scs", 1), ("scs", 4096)):
        prob = hostapi.Problem("irregular", n, n, n, fmt=fmt, Cc=64, sigma=sigma)
        y = gpu_spmv(gpu, prob, x)
        assert np.array_equal(y, yref), (fmt, sigma)  # bit-equal to the oracle's CRS loop at full size
        if fmt == "crs":  # symmetry of the operator as the GPU applies it
            w = gpu_spmv(gpu, prob, z)
            a, b = float(np.dot(z, y)), float(np.dot(x, w))
            assert abs(a - b) <= 1e-9 * max(abs(a), abs(b))
        for fused in (True, False):
            cg = hostapi.CG(prob, fused=fused)
            k = cg.solve(31, 0.0)
            hist[(fmt, sigma, fused)] = cg.history()
            assert k == 31
            cg.free()
        prob.free()
    keys = list(hist)
    for kx in keys[1:]:
        if kx[1] == 1:  # CRS and Sell-C-1, both loop forms: the same bits (same row order, same dot order)
            assert np.array_equal(hist[kx][0], hist[keys[0]][0]) and np.array_equal(hist[kx][1], hist[keys[0]][1]), kx
        else:  # sigma > 1: vectors in permuted row order, the fixed-order dots add in another order
            assert np.max(np.abs(hist[kx][0] - hist[keys[0]][0]) / hist[keys[0]][0]) < 1e-11, kx
    assert np.array_equal(hist[("scs", 4096, True)][0], hist[("scs", 4096, False)][0])
    o = po.cg(g, itermax=31, fmt="crs", dot="tree")  # ... and they are the oracle's
    assert np.array_equal(hist[keys[0]][0], o["rr"]) and np.array_equal(hist[keys[0]][1], o["pAp"])
    rr = hist[keys[0]][0]
    assert np.all(np.diff(np.log(rr)) < 0.5) and rr[-1] < 1e-3 * rr[0]  # converging (1.5e6 -> ~30 in 30 iterations)
    # the REFERENCE ITSELF at this size -- its own reader, convertMatrix and solveCG on the stand-in exported as
    # a 94 M-entry .mtx (tests/golden/make_golden_irregular_ref.py, build container; the oracle's sequential-dot run equals it bit
    # for bit, tests/test_oracle_pinning.py).  The GPU's history is within north_star's 1e-12 of it (observed 8.8e-13 / 9.7e-13: the
    # reference's sequential sum over 1.5 M elements is that far from a CG with exactly rounded dots, the GPU's order 3.4e-15).
    import json
    e = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cg_hist_irregular_ref.json")))["irregular80"]
    assert e["rows"] == g.nr and e["nnz"] == g.nnzTrue
    ref_rr, ref_pap = np.array([float(v) for v in e["rr"]]), np.array([float(v) for v in e["pAp"]])
    pap = hist[keys[0]][1]
    assert (np.abs(rr - ref_rr[:len(rr)]) / ref_rr[:len(rr)]).max() <= 1e-12
    assert (np.abs(pap - ref_pap[:len(pap)]) / ref_pap[:len(pap)]).max() <= 1e-12
