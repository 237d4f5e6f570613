"""configs[4] stand-in (host/sbh_irregular.c): the committed irregular SPD FE-like generator that replaces
SuiteSparse Flan_1565 (not available offline).  CPU only: properties the workload is specified by --
3x3-block rows, row lengths spread > 4x, a few % far couplings, symmetric, strictly diagonally dominant,
deterministic, and the ranks' row slices tile the one-rank matrix exactly."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import pyoracle as po
from sparsebench_amd import hostapi


def csr(n, rank=0, size=1):
    p = hostapi.Problem("irregular", n, n, n, fmt="crs", rank=rank, size=size, upload=False) if size == 1 else None
    assert p is not None
    rp = p.array("rowPtr").copy()
    col, val = p.gm_entries()
    meta = dict(nr=p.nr, nc=p.nc, nnz=p.nnz, nnzTrue=p.nnzTrue, totalNr=p.totalNr, totalNnz=p.totalNnz)
    b, xe = p.rhs()
    p.free()
    return rp, col, val, meta, b, xe


def test_structure_and_spd_properties():
    n = 20
    rp, col, val, meta, b, xe = csr(n)
    nr = 3 * n ** 3
    assert meta["nr"] == meta["totalNr"] == nr and meta["nnz"] == meta["totalNnz"] == meta["nnzTrue"] == rp[-1]
    assert np.all(b == 1.0) and xe is None  # enters like a file: b = 1 (src/CGSolver.c:33-35)
    lens = np.diff(rp.astype(np.int64))
    assert np.all(lens % 3 == 0) and lens.min() >= 3           # whole 3x3 blocks
    assert np.all(lens[0::3] == lens[1::3]) and np.all(lens[0::3] == lens[2::3])  # the 3 rows of a node share a pattern
    p5, p95 = np.percentile(lens, [5, 95])
    assert p95 / p5 >= 3.0 and lens.max() / max(lens.min(), 1) >= 4.0, (p5, p95, lens.min(), lens.max())
    A = sp.csr_matrix((val, col.astype(np.int64), rp.astype(np.int64)), shape=(nr, nr))
    assert A.has_sorted_indices or np.all(np.diff(A.indices[rp[5]:rp[6]]) > 0)
    for r in range(0, nr, 97):  # ascending, duplicate-free columns
        assert np.all(np.diff(col[rp[r]:rp[r + 1]].astype(np.int64)) > 0)
    assert abs(A - A.T).max() == 0.0  # exactly symmetric
    d = A.diagonal()
    off = abs(A).sum(axis=1).A1 - abs(d)
    assert np.all(d > 0) and np.all(d - off == 0.0625)  # strict dominance, exact arithmetic (multiples of 2^-21)
    rows = np.repeat(np.arange(nr), lens)
    far = np.abs(col.astype(np.int64) - rows) > 3 * (n * n + n + 2)
    assert 0.02 <= far.mean() <= 0.08, far.mean()  # "a few %" far couplings
    assert len(np.unique(val)) > 50000  # no value dictionary applies (unlike a stencil)


def test_deterministic_and_rank_slices_tile_the_matrix():
    n = 10
    rp, col, val, meta, _, _ = csr(n)
    rp2, col2, val2, _, _, _ = csr(n)
    assert np.array_equal(rp, rp2) and np.array_equal(col, col2) and np.array_equal(val, val2)
    nr = meta["nr"]
    for size in (2, 3):
        at = 0
        for rank in range(size):
            H = hostapi.host()
            # upload=False + size > 1 would need a setup exchange for commPartition: take the generator alone
            import ctypes as C

            class GM(C.Structure):
                _fields_ = [("nr", C.c_uint), ("nc", C.c_uint), ("nnz", C.c_uint), ("totalNr", C.c_uint), ("totalNnz", C.c_uint),
                            ("startRow", C.c_uint), ("stopRow", C.c_uint), ("rowPtr", C.POINTER(C.c_uint)), ("entries", C.c_void_p)]

            class Par(C.Structure):
                _fields_ = [("filename", C.c_char_p), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("itermax", C.c_int),
                            ("eps", C.c_double)]
            g, par = GM(), Par(b"irregular", n, n, n, 10, 0.0)
            H.sbh_matrix_generate_irregular(C.byref(g), C.byref(par), rank, size)
            assert g.startRow == at and g.totalNr == nr and g.totalNnz == meta["totalNnz"]
            lrp = np.ctypeslib.as_array(g.rowPtr, shape=(g.nr + 1,)).astype(np.int64)
            ent = np.ctypeslib.as_array(C.cast(g.entries, C.POINTER(C.c_ubyte)), shape=(int(lrp[-1]) * 16,))
            rec = ent.view(np.dtype([("col", "<u4"), ("pad", "<u4"), ("val", "<f8")]))
            lo, hi = int(rp[at]), int(rp[at + g.nr])
            assert np.array_equal(lrp + lo, rp[at:at + g.nr + 1].astype(np.int64))
            assert np.array_equal(rec["col"], col[lo:hi]) and np.array_equal(rec["val"], val[lo:hi])  # GLOBAL column ids
            at += g.nr
        assert at == nr


def test_oracle_cg_converges_on_it_and_formats_agree():
    """SPD in practice: CG (oracle, CPU) reduces the residual monotonically in the A-norm sense and the CRS and
    Sell-C-sigma (sigma = 1) restatements give the same history bit for bit (same per-row order, padding adds
    +0.0 * x[0]); with sigma > 1 the vectors live in permuted order, the fixed-order dots differ by rounding"""
    n = 8
    rp, col, val, meta, _, _ = csr(n)
    g = po.GMatrix.from_csr(rp, col, val, nc=meta["nc"])
    a = po.cg(g, itermax=60, fmt="crs", dot="tree")
    b = po.cg(g, itermax=60, fmt="scs", Cc=64, sigma=1, dot="tree")
    assert a["k"] == b["k"] == 60 and np.array_equal(a["rr"], b["rr"]) and np.array_equal(a["pAp"], b["pAp"])
    c = po.cg(g, itermax=60, fmt="scs", Cc=64, sigma=512, dot="tree")  # rows permuted: the dots add in another order
    assert c["k"] == 60 and np.max(np.abs(c["rr"] - a["rr"]) / a["rr"]) < 1e-11
    assert np.all(a["pAp"] > 0)  # p.Ap > 0 for every search direction: positive definite on the Krylov space
    assert a["rr"][-1] < 1e-12 * a["rr"][0] and np.all(np.isfinite(a["rr"])) and a["rr"][-1] > 0
