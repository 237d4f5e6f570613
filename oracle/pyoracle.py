"""ctypes front-end for the CPU oracle (liboracle.so) and, when present, the
reference's own code compiled in place (oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from sparsebench_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_u32p = C.POINTER(C.c_uint32)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)


class GMatrixS(C.Structure):
    _fields_ = [("nr", C.c_uint32), ("nc", C.c_uint32), ("nnz", C.c_uint32),
                ("nnzTrue", C.c_uint32), ("totalNr", C.c_uint32), ("totalNnz", C.c_uint32),
                ("startRow", C.c_uint32), ("stopRow", C.c_uint32), ("generated", C.c_int),
                ("rowPtr", _u32p), ("col", _u32p), ("val", _f64p)]


class PlanS(C.Structure):
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("externalCount", C.c_int),
                ("totalSendCount", C.c_int), ("indegree", C.c_int), ("outdegree", C.c_int),
                ("sources", _i32p), ("recvCounts", _i32p), ("rdispls", _i32p),
                ("destinations", _i32p), ("sendCounts", _i32p), ("sdispls", _i32p),
                ("elementsToSend", _i32p), ("externalGlobal", _u32p)]


class ScsS(C.Structure):
    _fields_ = [("nr", C.c_uint32), ("nc", C.c_uint32), ("nnz", C.c_uint32), ("C", C.c_uint32),
                ("sigma", C.c_uint32), ("nrPadded", C.c_uint32), ("nChunks", C.c_uint32),
                ("nElems", C.c_uint32), ("chunkPtr", _u32p), ("chunkLens", _u32p),
                ("colInd", _u32p), ("val", _f64p), ("oldToNewPerm", _u32p),
                ("newToOldPerm", _u32p)]


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "sb_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_src = os.environ.get("SB_REFERENCE", "/root/reference")
    if os.path.isdir(os.path.join(ref_src, "src")):
        if force or not os.path.exists(os.path.join(HERE, "_ref", "libsbref_crs.so")):
            subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")],
                                  stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(os.path.join(HERE, "liboracle.so"))
    P = C.POINTER
    L.orc_generate.restype = P(GMatrixS)
    L.orc_generate.argtypes = [C.c_int] * 6
    L.orc_mm_load.restype = P(GMatrixS)
    L.orc_mm_load.argtypes = [C.c_char_p]
    L.orc_mm_load_part.restype = P(GMatrixS)
    L.orc_mm_load_part.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.orc_gm_from_arrays.restype = P(GMatrixS)
    L.orc_gm_from_arrays.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_gm_free.argtypes = [P(GMatrixS)]
    L.orc_partition.restype = P(PlanS)
    L.orc_partition.argtypes = [P(P(GMatrixS)), C.c_int]
    L.orc_plan_free.argtypes = [P(PlanS), C.c_int]
    L.orc_convert_scs.restype = P(ScsS)
    L.orc_convert_scs.argtypes = [P(GMatrixS), C.c_uint32, C.c_uint32]
    L.orc_scs_free.argtypes = [P(ScsS)]
    L.orc_spmv_crs.argtypes = [P(GMatrixS), C.c_void_p, C.c_void_p]
    L.orc_spmv_scs.argtypes = [P(ScsS), C.c_void_p, C.c_void_p]
    L.orc_spmv_scs_literal.argtypes = [P(ScsS), C.c_void_p, C.c_void_p]
    L.orc_waxpby.argtypes = [C.c_uint32, C.c_double, C.c_void_p, C.c_double, C.c_void_p,
                             C.c_void_p]
    L.orc_ddot_seq.restype = C.c_double
    L.orc_ddot_seq.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    L.orc_ddot_tree.restype = C.c_double
    L.orc_ddot_tree.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    L.orc_ddot_exact.restype = C.c_double
    L.orc_ddot_exact.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    L.orc_ddot_partials.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_reduce_final.restype = C.c_double
    L.orc_reduce_final.argtypes = [C.c_uint32, C.c_void_p]
    L.orc_cg.restype = C.c_int
    L.orc_cg.argtypes = [P(P(GMatrixS)), P(PlanS), C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                         C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, P(C.c_int),
                         C.c_void_p, P(C.c_int), P(_f64p), P(C.c_double)]
    L.orc_time_cg_iters.restype = C.c_double
    L.orc_time_cg_iters.argtypes = [P(GMatrixS), C.c_int, P(C.c_int)]
    L.orc_time_spmv.restype = C.c_double
    L.orc_time_spmv.argtypes = [P(GMatrixS), C.c_int, P(C.c_int)]
    _lib = L
    return L


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class GMatrix:
    """Local general matrix owned by the oracle library."""

    def __init__(self, ptr):
        self.ptr = ptr
        self.s = ptr.contents

    @classmethod
    def generate(cls, nx, ny, nz, rank=0, size=1, use7pt=False):
        return cls(lib().orc_generate(nx, ny, nz, rank, size, int(use7pt)))

    @classmethod
    def from_mtx(cls, path, rank=0, size=1):
        return cls(lib().orc_mm_load_part(os.fsencode(path), rank, size))

    @classmethod
    def from_csr(cls, rowPtr, col, val, nc=None):
        rowPtr = np.ascontiguousarray(rowPtr, dtype=np.uint32)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        nr = len(rowPtr) - 1
        return cls(lib().orc_gm_from_arrays(nr, nc if nc is not None else nr, _p(rowPtr),
                                            _p(col), _p(val)))

    nr = property(lambda self: self.s.nr)
    nc = property(lambda self: self.s.nc)
    nnz = property(lambda self: self.s.nnz)
    nnzTrue = property(lambda self: self.s.nnzTrue)
    totalNr = property(lambda self: self.s.totalNr)
    totalNnz = property(lambda self: self.s.totalNnz)
    startRow = property(lambda self: self.s.startRow)
    stopRow = property(lambda self: self.s.stopRow)
    generated = property(lambda self: bool(self.s.generated))

    @property
    def rowPtr(self):
        return _arr(self.s.rowPtr, self.nr + 1, np.uint32)

    @property
    def col(self):
        return _arr(self.s.col, self.nnzTrue, np.uint32)

    @property
    def val(self):
        return _arr(self.s.val, self.nnzTrue, np.float64)

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert len(x) >= self.nc
        y = np.empty(self.nr, dtype=np.float64)
        lib().orc_spmv_crs(self.ptr, _p(x), _p(y))
        return y

    def to_scs(self, Cc, sigma):
        return Scs(lib().orc_convert_scs(self.ptr, Cc, sigma))

    def rhs(self):
        """b of initVectors (src/CGSolver.c:25-36)."""
        if self.generated:
            return 27.0 - (np.diff(self.rowPtr.astype(np.int64)) - 1).astype(np.float64)
        return np.ones(self.nr)

    def free(self):
        if self.ptr:
            lib().orc_gm_free(self.ptr)
            self.ptr = None


class Scs:
    def __init__(self, ptr):
        self.ptr = ptr
        self.s = ptr.contents

    def __getattr__(self, k):
        s = object.__getattribute__(self, "s")
        if k in ("nr", "nc", "nnz", "C", "sigma", "nrPadded", "nChunks", "nElems"):
            return getattr(s, k)
        sizes = {"chunkPtr": s.nChunks + 1, "chunkLens": s.nChunks, "colInd": s.nElems,
                 "oldToNewPerm": s.nr, "newToOldPerm": s.nr}
        if k in sizes:
            return _arr(getattr(s, k), sizes[k], np.uint32)
        if k == "val":
            return _arr(s.val, s.nElems, np.float64)
        raise AttributeError(k)

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.nr, dtype=np.float64)
        lib().orc_spmv_scs(self.ptr, _p(x), _p(y))
        return y

    def spmv_literal(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.nrPadded, dtype=np.float64)
        lib().orc_spmv_scs_literal(self.ptr, _p(x), _p(y))
        return y

    def free(self):
        if self.ptr:
            lib().orc_scs_free(self.ptr)
            self.ptr = None


class Plans:
    """Halo plans of P ranks (orc_partition rewrites the matrices' columns)."""

    def __init__(self, locals_):
        self.P = len(locals_)
        self.locals = locals_
        arr = (C.POINTER(GMatrixS) * self.P)(*[g.ptr for g in locals_])
        self._arr = arr
        self.ptr = lib().orc_partition(arr, self.P)

    def plan(self, r):
        s = self.ptr[r]
        return {
            "externalCount": s.externalCount, "totalSendCount": s.totalSendCount,
            "indegree": s.indegree, "outdegree": s.outdegree,
            "sources": _arr(s.sources, s.indegree, np.int32).copy(),
            "recvCounts": _arr(s.recvCounts, s.indegree, np.int32).copy(),
            "rdispls": _arr(s.rdispls, s.indegree, np.int32).copy(),
            "destinations": _arr(s.destinations, s.outdegree, np.int32).copy(),
            "sendCounts": _arr(s.sendCounts, s.outdegree, np.int32).copy(),
            "sdispls": _arr(s.sdispls, s.outdegree, np.int32).copy(),
            "elementsToSend": _arr(s.elementsToSend, s.totalSendCount, np.int32).copy(),
            "externalGlobal": _arr(s.externalGlobal, s.externalCount, np.uint32).copy(),
        }


def waxpby(alpha, x, beta, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    w = np.empty_like(x)
    lib().orc_waxpby(len(x), alpha, _p(x), beta, _p(y), _p(w))
    return w


def ddot_seq(x, y):
    return lib().orc_ddot_seq(len(x), _p(x), _p(y))


def ddot_tree(x, y):
    return lib().orc_ddot_tree(len(x), _p(x), _p(y))


def ddot_exact(x, y):
    return lib().orc_ddot_exact(len(x), _p(x), _p(y))


def ddot_partials(x, y):
    q = np.empty((len(x) + 255) // 256, dtype=np.float64)
    lib().orc_ddot_partials(len(x), _p(x), _p(y), _p(q))
    return q


def reduce_final(q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    return lib().orc_reduce_final(len(q), _p(q))


def cg(locals_, plans=None, fmt="crs", Cc=64, sigma=1, itermax=150, eps=0.0, dot="seq",
       rank_sum="order", want_x=False):
    """Run the oracle CG.  Returns dict(k, rr, pAp, max_err[, x])."""
    if isinstance(locals_, GMatrix):
        locals_ = [locals_]
    Pn = len(locals_)
    arr = (C.POINTER(GMatrixS) * Pn)(*[g.ptr for g in locals_])
    rr = np.zeros(itermax + 2)
    pap = np.zeros(itermax + 2)
    nrr, npap = C.c_int(0), C.c_int(0)
    err = C.c_double(0.0)
    xo = (_f64p * Pn)() if want_x else None
    k = lib().orc_cg(arr, plans.ptr if plans is not None else None, Pn,
                     0 if fmt == "crs" else 1, Cc, sigma, itermax, eps,
                     {"seq": 0, "tree": 1, "exact": 2}[dot], 0 if rank_sum == "order" else 1, _p(rr),
                     C.byref(nrr), _p(pap), C.byref(npap), xo, C.byref(err))
    out = {"k": k, "rr": rr[:nrr.value].copy(), "pAp": pap[:npap.value].copy(),
           "max_err": err.value}
    if want_x:
        out["x"] = [_arr(xo[q], locals_[q].nr, np.float64).copy() for q in range(Pn)]
    return out


# ---------------------------------------------------------------------------
# The reference itself, compiled in place (oracle/_ref), if it was built.
# ---------------------------------------------------------------------------
def ref_available(kind="crs"):
    return os.path.exists(os.path.join(HERE, "_ref", "libsbref_%s.so" % kind))


class Ref:
    """One loaded copy of a reference build (global state inside: one at a time)."""

    def __init__(self, kind="crs"):
        path = os.path.join(HERE, "_ref", "libsbref_%s.so" % kind)
        self.kind = kind
        # the reference exports the same names as the product's drop-in libraries
        # (matrixGenerate, convertMatrix, ...): bind its calls to its own definitions
        L = C.CDLL(path, mode=os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
        L.sbref_setup.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.sbref_hist_val.restype = C.c_double
        L.sbref_rowptr.restype = _u32p
        L.sbref_entries.argtypes = [C.c_void_p, C.c_void_p]
        L.sbref_spmv.argtypes = [C.c_void_p, C.c_void_p]
        L.sbref_waxpby.argtypes = [C.c_uint, C.c_double, C.c_void_p, C.c_double, C.c_void_p,
                                   C.c_void_p]
        L.sbref_ddot.restype = C.c_double
        L.sbref_ddot.argtypes = [C.c_uint, C.c_void_p, C.c_void_p]
        L.sbref_solve_cg.argtypes = [C.c_int, C.c_double]
        if kind.startswith("scs"):
            L.sbref_scs_field.restype = C.c_uint
            L.sbref_scs_array.restype = _u32p
            L.sbref_scs_val.restype = _f64p
        self.L = L

    def setup(self, filename="generate", nx=8, ny=8, nz=8, Cc=1, sigma=1):
        self.L.sbref_setup(os.fsencode(filename), nx, ny, nz, Cc, sigma)
        self.nr, self.nc = self.L.sbref_nr(), self.L.sbref_nc()
        self.nnz, self.nnzTrue = self.L.sbref_nnz(), self.L.sbref_nnz_true()
        self.totalNr, self.totalNnz = self.L.sbref_total_nr(), self.L.sbref_total_nnz()

    def csr(self):
        rp = _arr(self.L.sbref_rowptr(), self.nr + 1, np.uint32).copy()
        col = np.empty(self.nnzTrue, dtype=np.uint32)
        val = np.empty(self.nnzTrue, dtype=np.float64)
        self.L.sbref_entries(_p(col), _p(val))
        return rp, col, val

    def spmv(self, x, ny=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(ny if ny is not None else self.nr, dtype=np.float64)
        self.L.sbref_spmv(_p(x), _p(y))
        return y

    def waxpby(self, a, x, b, y):
        w = np.empty_like(x)
        self.L.sbref_waxpby(len(x), a, _p(x), b, _p(y), _p(w))
        return w

    def ddot(self, x, y):
        return self.L.sbref_ddot(len(x), _p(x), _p(y))

    def solve_cg(self, itermax=150, eps=0.0):
        k = self.L.sbref_solve_cg(itermax, eps)
        n = self.L.sbref_hist_len()
        vals = np.array([self.L.sbref_hist_val(i) for i in range(n)])
        kinds = np.array([self.L.sbref_hist_kind(i) for i in range(n)])
        return {"k": k, "rr": vals[kinds == 0], "pAp": vals[kinds == 1]}

    def scs(self):
        f = self.L.sbref_scs_field
        d = {"C": f(0), "sigma": f(1), "nChunks": f(2), "nrPadded": f(3), "nElems": f(4)}
        a = self.L.sbref_scs_array
        d["chunkPtr"] = _arr(a(0), d["nChunks"] + 1, np.uint32).copy()
        d["chunkLens"] = _arr(a(1), d["nChunks"], np.uint32).copy()
        d["colInd"] = _arr(a(2), d["nElems"], np.uint32).copy()
        d["oldToNewPerm"] = _arr(a(3), self.nr, np.uint32).copy()
        d["newToOldPerm"] = _arr(a(4), self.nr, np.uint32).copy()
        d["val"] = _arr(self.L.sbref_scs_val(), d["nElems"], np.float64).copy()
        return d
