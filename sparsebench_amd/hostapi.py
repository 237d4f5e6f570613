"""Python mirror of the C host side (libsparsebench_host.so) for tests and bench.py.

`Problem` sequences what the C driver does -- matrixGenerate | MMMatrixRead ->
commPartition -> convertMatrix -- through the flat surface in host/sbh_flat.c, and
`CG` wraps the HIP layer's solveCG (sb_cg_*).  Plumbing only: no arithmetic happens in
Python, and nothing here falls back to the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import capi

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(HERE, "lib", "libsparsebench_host.so")
vp = C.c_void_p
_host = None

# setup-exchange callbacks (include/sparsebench/sparsebench.h: sbh_exchange)
ALLGATHER_FN = C.CFUNCTYPE(None, vp, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int))
ALLTOALLV_FN = C.CFUNCTYPE(None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))


class ExchangeS(C.Structure):
    _fields_ = [("ctx", vp), ("allgather_ints", ALLGATHER_FN), ("alltoallv_ints", ALLTOALLV_FN)]


def host():
    global _host
    if _host is not None:
        return _host
    capi.load()  # same libsbhip.so instance the host library links against
    if not os.path.exists(HOST_LIB):
        raise RuntimeError("sparsebench_amd: %s is missing -- run `make host`" % HOST_LIB)
    H = C.CDLL(HOST_LIB)
    H.sbh_problem_create.restype = vp
    H.sbh_problem_create.argtypes = [C.c_char_p] + [C.c_int] * 9
    H.sbh_problem_matrix.restype = vp
    H.sbh_problem_matrix.argtypes = [vp]
    H.sbh_problem_halo.restype = vp
    H.sbh_problem_halo.argtypes = [vp]
    H.sbh_problem_setup_seconds.restype = C.c_double
    H.sbh_problem_setup_seconds.argtypes = [vp]
    H.sbh_problem_scalar.restype = C.c_uint
    H.sbh_problem_scalar.argtypes = [vp, C.c_int]
    H.sbh_problem_array.restype = vp
    H.sbh_problem_array.argtypes = [vp, C.c_int]
    H.sbh_problem_values.restype = vp
    H.sbh_problem_values.argtypes = [vp]
    H.sbh_problem_gm_entries.argtypes = [vp, vp, vp]
    H.sbh_convert_mtx_to_bmx.restype = None
    H.sbh_convert_mtx_to_bmx.argtypes = [C.c_char_p]
    H.sbh_problem_rhs.restype = C.c_int
    H.sbh_problem_rhs.argtypes = [vp, vp, vp]
    H.sbh_problem_free.argtypes = [vp]
    H.commSetExchange.argtypes = [vp]
    H.MMMatrixRead.argtypes = [vp, C.c_char_p]
    H.commDistributeMatrix.argtypes = [vp, vp, vp]
    H.sbh_exchange_rccl.restype = vp
    _host = H
    return H


_SCALARS = ["nr", "nc", "nnz", "nnzTrue", "totalNr", "totalNnz", "startRow", "stopRow", "C",
            "sigma", "nChunks", "nrPadded", "nElems", "externalCount", "totalSendCount",
            "indegree", "outdegree"]
_ARRAYS = {"rowPtr": 0, "rowNnz": 1, "crs_colInd": 2, "chunkPtr": 3, "chunkLens": 4,
           "scs_colInd": 5, "oldToNewPerm": 6, "newToOldPerm": 7, "elementsToSend": 8,
           "sources": 9, "recvCounts": 10, "rdispls": 11, "destinations": 12, "sendCounts": 13,
           "sdispls": 14, "externalGlobal": 15}


def convert_mtx_to_bmx(path):
    """file.mtx -> file.bmx next to it (the driver's -c option); SB_BMX_FP64=1 keeps fp64 values"""
    host().sbh_convert_mtx_to_bmx(os.fspath(path).encode())
    return os.path.splitext(os.fspath(path))[0] + ".bmx"


def read_bmx_slice(path, rank=0, size=1):
    """rank's row slice of a .bmx file as (startRow, rowPtr, col, val), global column ids"""
    H = host()
    H.sbh_bmx_read.restype = vp
    H.sbh_bmx_read.argtypes = [C.c_char_p, C.c_int, C.c_int]
    H.sbh_gm_scalar.restype = C.c_uint
    H.sbh_gm_scalar.argtypes = [vp, C.c_int]
    H.sbh_gm_copy.argtypes = [vp, vp, vp, vp]
    H.sbh_gm_free.argtypes = [vp]
    g = H.sbh_bmx_read(os.fspath(path).encode(), rank, size)
    nr, nnz, start = (H.sbh_gm_scalar(g, i) for i in range(3))
    rp, col, val = np.empty(nr + 1, np.uint32), np.empty(nnz, np.uint32), np.empty(nnz, np.float64)
    H.sbh_gm_copy(g, rp.ctypes.data_as(vp), col.ctypes.data_as(vp), val.ctypes.data_as(vp))
    H.sbh_gm_free(g)
    return start, rp, col, val


def _view(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    ct = {np.uint32: C.c_uint32, np.int32: C.c_int32, np.float64: C.c_double}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(int(n),))


class Problem:
    """One rank's matrix: generated HPCG stencil or a .mtx file, partitioned, converted
    and (upload=True) resident in HBM."""

    def __init__(self, filename="generate", nx=16, ny=16, nz=16, fmt="scs", Cc=64, sigma=1,
                 rank=0, size=1, upload=True):
        self.H = host()
        self.fmt = fmt
        self.upload = upload
        self.ptr = self.H.sbh_problem_create(os.fsencode(filename), nx, ny, nz,
                                             0 if fmt == "crs" else 1, Cc, sigma, rank, size,
                                             1 if upload else 0)
        for i, name in enumerate(_SCALARS):
            setattr(self, name, int(self.H.sbh_problem_scalar(self.ptr, i)))

    def array(self, name):
        n = {"rowPtr": self.nr + 1, "rowNnz": self.nr, "crs_colInd": self.nnzTrue,
             "chunkPtr": self.nChunks + 1, "chunkLens": self.nChunks, "scs_colInd": self.nElems,
             "oldToNewPerm": self.nr, "newToOldPerm": self.nr,
             "elementsToSend": self.totalSendCount, "sources": self.indegree,
             "recvCounts": self.indegree, "rdispls": self.indegree,
             "destinations": self.outdegree, "sendCounts": self.outdegree,
             "sdispls": self.outdegree, "externalGlobal": self.externalCount}[name]
        dt = np.int32 if name in ("elementsToSend", "sources", "recvCounts", "rdispls",
                                  "destinations", "sendCounts", "sdispls") else np.uint32
        return _view(self.H.sbh_problem_array(self.ptr, _ARRAYS[name]), n, dt)

    def values(self):
        n = self.nnzTrue if self.fmt == "crs" else self.nElems
        return _view(self.H.sbh_problem_values(self.ptr), n, np.float64)

    def gm_entries(self):
        col = np.empty(self.nnzTrue, dtype=np.uint32)
        val = np.empty(self.nnzTrue, dtype=np.float64)
        self.H.sbh_problem_gm_entries(self.ptr, col.ctypes.data_as(vp), val.ctypes.data_as(vp))
        return col, val

    def rhs(self):
        b = np.empty(self.nr)
        xe = np.empty(self.nr)
        gen = self.H.sbh_problem_rhs(self.ptr, b.ctypes.data_as(vp), xe.ctypes.data_as(vp))
        return b, (xe if gen else None)

    @property
    def matrix(self):
        return self.H.sbh_problem_matrix(self.ptr)

    @property
    def halo(self):
        return self.H.sbh_problem_halo(self.ptr)

    @property
    def setup_seconds(self):
        return self.H.sbh_problem_setup_seconds(self.ptr)

    def spmv_bytes(self):
        """algorithmic bytes of one SpMV in the reference's layout"""
        return capi.load().sb_matrix_spmv_bytes(self.matrix)

    def stream_bytes(self):
        """bytes the selected kernel really moves (compressed mirror, if in use)"""
        return capi.load().sb_matrix_stream_bytes(self.matrix)

    def use_packed(self, mode):
        capi.load().sb_matrix_use_packed(self.matrix, int(mode))
        return capi.load().sb_matrix_packed_mode(self.matrix)

    def placement_report(self):
        """what the upload's placement tuner saw (us of a proxy loop body: p update | SpMV on the reference-layout stream | r update):
        with the first vectors' arena tried and the stream where hipMalloc put it, at the pair kept, at the slowest pair; None if
        it did not run (SB_PLACE=0, or a stream below 64 MB)"""
        us = (C.c_float * 3)()
        n = capi.load().sb_matrix_placement_report(self.matrix, us)
        return {"probes_timed": n, "us_first_pair": round(us[0], 2), "us_kept": round(us[1], 2), "us_slowest": round(us[2], 2)} if n else None

    def pack_info(self):
        L = capi.load()
        uni = C.c_uint32(0)
        pats = L.sb_matrix_row_patterns(self.matrix, C.byref(uni))
        return {"level": L.sb_matrix_pack_level(self.matrix), "mode": L.sb_matrix_packed_mode(self.matrix),
                "row_patterns": pats, "uniform_chunks": uni.value,
                "lds_window_doubles": L.sb_matrix_lds_window(self.matrix),
                "pattern_classes": L.sb_matrix_pattern_classes(self.matrix)}

    def free(self):
        if self.ptr:
            self.H.sbh_problem_free(self.ptr)
            self.ptr = None


class CG:
    """solveCG on the GPU (sb_cg_*): state in HBM, loop without host round trips."""

    def __init__(self, problem, fused=True, graph=False, fuse_p=-1, fuse_alpha=-1, fuse_beta=-1):
        self.L = capi.load()
        self.problem = problem
        b, xe = problem.rhs()
        self.ptr = self.L.sb_cg_create(problem.matrix, problem.halo, b.ctypes.data_as(vp),
                                       xe.ctypes.data_as(vp) if xe is not None else None)
        # fused: True = the default (1: dots fused into their producers), False = the reference's op list, or the
        # level itself (0, 1, 2; 2 = additionally the vector phase as one launch where possible)
        self.L.sb_cg_set_fused(self.ptr, int(fused))
        self.L.sb_cg_set_graph(self.ptr, int(graph))
        self.L.sb_cg_set_fuse_p(self.ptr, int(fuse_p))  # -1: default; 1 / 0: the p update inside the SpMV where possible / not
        self.L.sb_cg_set_fuse_alpha(self.ptr, int(fuse_alpha))  # -1: default; 1 / 0: the alpha step inside the r update (one rank) / not
        self.L.sb_cg_set_fuse_beta(self.ptr, int(fuse_beta))  # the beta step at the head of the p update where that is a launch of its own
        self.itermax = 0

    def vector_phase(self):
        """spans per wave of the one-launch vector phase, 0 if the solver uses the separate launches"""
        return self.L.sb_cg_vector_phase(self.ptr)

    def launches_per_body(self):
        return self.L.sb_cg_launches_per_body(self.ptr)

    def fuse_p(self):
        """1: the loop takes the p update inside the SpMV launch (4 launches per body)"""
        return self.L.sb_cg_fuse_p(self.ptr)

    def collectives_per_body(self):
        return self.L.sb_cg_collectives_per_body(self.ptr)

    def solve(self, itermax=150, eps=0.0):
        self.itermax = itermax
        return self.L.sb_cg_solve(self.ptr, itermax, eps)

    def start(self, itermax, eps=0.0):
        """prologue only; follow with run_iters() and finish()"""
        self.itermax = itermax
        self.L.sb_cg_start(self.ptr, itermax, eps)

    def run_iters(self, iters):
        self.L.sb_cg_run_iters(self.ptr, iters)

    def finish(self):
        return self.L.sb_cg_finish(self.ptr)

    def history(self):
        cap = self.itermax + 2
        rr = np.zeros(cap)
        pap = np.zeros(cap)
        npap = C.c_int(0)
        nrr = self.L.sb_cg_history(self.ptr, rr.ctypes.data_as(vp), cap, pap.ctypes.data_as(vp),
                                   cap, C.byref(npap))
        return rr[:nrr].copy(), pap[:npap.value].copy()

    def solution(self):
        x = np.empty(self.problem.nr)
        self.L.sb_cg_solution(self.ptr, x.ctypes.data_as(vp))
        return x

    def check_residual(self):
        return self.L.sb_cg_check_residual(self.ptr)

    def counters(self):
        out = (C.c_int * 5)()
        self.L.sb_cg_counters(self.ptr, out)
        return dict(zip(["stop", "stop_next", "iters", "n_rr", "n_pAp"], list(out)))

    def spmv_timing(self, on):
        self.L.sb_cg_spmv_timing(self.ptr, int(on))

    def spmv_ms(self):
        n = C.c_int(0)
        ms = self.L.sb_cg_spmv_ms(self.ptr, C.byref(n))
        return ms, n.value

    PHASES = ("p_update", "halo", "spmv", "alpha_step", "r_update", "beta_step", "dot_pass")

    def phase_timing(self, on):
        self.L.sb_cg_phase_timing(self.ptr, int(on))

    def phase_us(self):
        """{phase: (mean microseconds per occurrence, occurrences)} since phase_timing(True)"""
        ms, cnt = (C.c_double * 8)(), (C.c_int * 8)()
        n = self.L.sb_cg_phase_ms(self.ptr, ms, cnt)
        return {self.PHASES[i]: (1e3 * ms[i] / cnt[i], cnt[i]) for i in range(n) if cnt[i]}

    def loop_ms(self):
        return self.L.sb_cg_loop_ms(self.ptr)

    def region_ms(self):
        out = np.zeros(4)
        self.L.sb_cg_region_ms(self.ptr, out.ctypes.data_as(vp))
        return dict(zip(["waxpby", "spMVM", "ddot", "comm"], out))

    def free(self):
        if self.ptr:
            self.L.sb_cg_free(self.ptr)
            self.ptr = None
