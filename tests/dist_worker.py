"""Worker of tests/test_dist_gloo.py: one process per rank, gloo backend, CPU only.

Exercises the N>1 path of the PRODUCT host code -- commPartition with a launcher-
provided setup exchange (here torch.distributed/gloo instead of RCCL) -- and checks
  1. the halo plan and the renumbered local matrix against the oracle's single-process
     partition of the same problem (bit for bit), and
  2. a CG run in which every rank uses the plan for its halo exchange and gloo for the
     dot all-reduce (CPU arithmetic by the oracle's kernels: this is the checker),
     against the history captured from the reference under mpiexec.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from irregular_locs import irregular_locs  # noqa: E402
from sparsebench_amd import hostapi  # noqa: E402


class MME(C.Structure):
    _fields_ = [("row", C.c_int), ("col", C.c_int), ("val", C.c_double)]


class MMM(C.Structure):
    _fields_ = [("count", C.c_size_t), ("nr", C.c_int), ("nnz", C.c_int), ("totalNr", C.c_int), ("totalNnz", C.c_int),
                ("startRow", C.c_int), ("stopRow", C.c_int), ("entries", C.POINTER(MME))]


class CommS(C.Structure):  # include/sparsebench/sparsebench.h: Comm
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("logFile", C.c_void_p), ("externalCount", C.c_int),
                ("totalSendCount", C.c_int), ("elementsToSend", C.c_void_p), ("indegree", C.c_int), ("outdegree", C.c_int),
                ("sources", C.c_void_p), ("recvCounts", C.c_void_p), ("rdispls", C.c_void_p), ("destinations", C.c_void_p),
                ("sendCounts", C.c_void_p), ("sdispls", C.c_void_p), ("sendBuffer", C.c_void_p), ("dev", C.c_void_p),
                ("externalGlobal", C.c_void_p)]


def distribute_case(H, rank, size):
    """commDistributeMatrix with the REFERENCE's contract (src/main.c:63-70, src/comm.c:311-402): only the master has
    read the file, the other ranks pass an uninitialised MMMatrix; every rank must end up with exactly the rows the
    file rule gives it (src/comm.c:35-38), entries in file order, and the global counts."""
    path = os.path.join(ROOT, "tests", "golden", "ref", "matrix_band_klein.mtx")
    comm = CommS()
    comm.rank, comm.size = rank, size
    mm, loc = MMM(), MMM()
    if rank == 0:
        H.MMMatrixRead(C.byref(mm), path.encode())
    else:  # garbage, as an uninitialised stack variable would be
        C.memset(C.byref(mm), 0x5A, C.sizeof(mm))
    H.commDistributeMatrix(C.byref(comm), C.byref(mm), C.byref(loc))
    g = po.GMatrix.from_mtx(path, rank, size)  # the oracle's row slice of the same file
    assert loc.nr == g.nr and loc.startRow == g.startRow and loc.stopRow == g.stopRow, (loc.nr, g.nr)
    assert loc.totalNr == 100 and loc.totalNnz == 298 and loc.count == g.nnzTrue == loc.nnz
    rows = np.repeat(np.arange(g.nr), np.diff(g.rowPtr.astype(np.int64))) + g.startRow
    got = [(loc.entries[i].row, loc.entries[i].col, loc.entries[i].val) for i in range(loc.count)]
    assert [e[0] for e in got] == rows.tolist()
    assert [e[1] for e in got] == g.col.tolist() and [e[2] for e in got] == g.val.tolist()
    dist.barrier()
    # Empty rows at the edges of a rank's range (and, with enough ranks, a rank without any entry): the rank's
    # startRow / stopRow stay its ROW RANGE (src/comm.c:35-38) -- a documented deviation from src/comm.c:385-387, which takes
    # the first / last entry's row and would shift the ownership of the empty rows; the scatter must deliver
    # exactly the entries of the range, in file order, also when a slice is empty.
    n = 13
    path2 = "/tmp/sb_empty_rows_%s.mtx" % os.environ.get("MASTER_PORT", "0")
    base, extra = n // size, n % size
    firsts = [r * base + min(r, extra) for r in range(size)]
    empty = set(firsts[1:]) | {f - 1 for f in firsts[1:]} | ({n - 1} if size > 2 else set())  # rows around every rank boundary
    if size >= 4:
        empty |= set(range(firsts[size - 2], firsts[size - 1]))  # the second-last rank gets no entry at all
    ents = [(i, j, 1.0 + i + 0.25 * j) for i in range(n) if i not in empty for j in (max(i - 1, 0), i, min(i + 1, n - 1)) if j == i or j != i]
    ents = sorted(set(ents))
    if rank == 0:
        with open(path2, "w") as f:
            f.write("%%MatrixMarket matrix coordinate real general\n")
            f.write("%d %d %d\n" % (n, n, len(ents)))
            for i, j, v in ents:
                f.write("%d %d %.17g\n" % (i + 1, j + 1, v))
    dist.barrier()
    mm2, loc2 = MMM(), MMM()
    if rank == 0:
        H.MMMatrixRead(C.byref(mm2), path2.encode())
    else:
        C.memset(C.byref(mm2), 0x5A, C.sizeof(mm2))
    H.commDistributeMatrix(C.byref(comm), C.byref(mm2), C.byref(loc2))
    lo = firsts[rank]
    hi = (firsts[rank + 1] if rank + 1 < size else n) - 1
    want = [e for e in ents if lo <= e[0] <= hi]
    assert (loc2.startRow, loc2.stopRow, loc2.nr) == (lo, hi, hi - lo + 1), (loc2.startRow, loc2.stopRow, lo, hi)
    assert loc2.totalNr == n and loc2.totalNnz == len(ents) and loc2.count == len(want) == loc2.nnz
    assert [(loc2.entries[i].row, loc2.entries[i].col, loc2.entries[i].val) for i in range(loc2.count)] == want
    if size >= 4:
        assert (len(want) == 0) == (rank == size - 2)
    dist.barrier()
    if rank == 0:
        os.unlink(path2)
        print("DIST_OK distribute", size, flush=True)
    dist.destroy_process_group()


def irregular_case(H, rank, size):
    """configs[4] stand-in on several ranks: generator slice -> commPartition (far couplings give every rank many
    neighbours) -> plan and renumbered matrix equal to the oracle's partition of the one-rank matrix's row slices"""
    n = 10
    prob = hostapi.Problem("irregular", n, n, n, fmt="crs", rank=rank, size=size, upload=False)
    locs = irregular_locs(n, size)  # the oracle partitions row slices of the ONE-rank matrix (global column ids)
    plans = po.Plans(locs)
    mine, g = plans.plan(rank), locs[rank]
    assert prob.nr == g.nr and prob.nc == g.nc, (prob.nc, g.nc)
    col, val = prob.gm_entries()
    assert np.array_equal(col, g.col) and np.array_equal(val, g.val)
    for f in ("externalCount", "totalSendCount", "indegree", "outdegree"):
        assert getattr(prob, f) == mine[f], f
    for f in ("sources", "recvCounts", "rdispls", "destinations", "sendCounts", "sdispls", "elementsToSend", "externalGlobal"):
        assert np.array_equal(prob.array(f), mine[f]), f
    assert prob.indegree == size - 1  # far couplings: everybody talks to everybody
    dist.barrier()
    if rank == 0:
        print("DIST_OK irregular", size, flush=True)
    dist.destroy_process_group()


def main():
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    H = hostapi.host()

    def allgather(ctx, mine, n, out):
        t = torch.tensor([mine[i] for i in range(n)], dtype=torch.int32)
        outs = [torch.zeros(n, dtype=torch.int32) for _ in range(size)]
        dist.all_gather(outs, t)
        flat = torch.cat(outs).tolist()
        for i, v in enumerate(flat):
            out[i] = v

    def alltoallv(ctx, sbuf, scnt, sdsp, rbuf, rcnt, rdsp):
        reqs = []
        recv = {}
        for r in range(size):
            if r == rank:
                continue
            if scnt[r]:
                t = torch.tensor([sbuf[sdsp[r] + i] for i in range(scnt[r])], dtype=torch.int32)
                reqs.append(dist.isend(t, r))
            if rcnt[r]:
                recv[r] = torch.zeros(rcnt[r], dtype=torch.int32)
                reqs.append(dist.irecv(recv[r], r))
        for q in reqs:
            q.wait()
        for r, t in recv.items():
            for i, v in enumerate(t.tolist()):
                rbuf[rdsp[r] + i] = v
        for i in range(scnt[rank]):
            rbuf[rdsp[rank] + i] = sbuf[sdsp[rank] + i]

    cb1, cb2 = hostapi.ALLGATHER_FN(allgather), hostapi.ALLTOALLV_FN(alltoallv)
    xchg = hostapi.ExchangeS(None, cb1, cb2)
    H.commSetExchange(C.byref(xchg))

    case = sys.argv[1]
    if case == "distribute":
        return distribute_case(H, rank, size)
    if case == "irregular":
        return irregular_case(H, rank, size)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "cg_hist_mpi.json")))
    if case == "hpcg":
        n = 16 if size <= 4 else 8
        key = "hpcg%d_x%d" % (n, size)
        prob = hostapi.Problem("generate", n, n, n, fmt="crs", rank=rank, size=size, upload=False)
        locs = [po.GMatrix.generate(n, n, n, r, size) for r in range(size)]
    else:
        key = "band_klein_x%d" % size
        path = os.path.join(ROOT, "tests", "golden", "ref", "matrix_band_klein.mtx")
        prob = hostapi.Problem(path, fmt="crs", rank=rank, size=size, upload=False)
        locs = [po.GMatrix.from_mtx(path, r, size) for r in range(size)]
    plans = po.Plans(locs)  # oracle: all ranks in one process
    mine, g = plans.plan(rank), locs[rank]

    # 1. plan + renumbered matrix == oracle
    assert prob.nr == g.nr and prob.nc == g.nc, (prob.nc, g.nc)
    col, val = prob.gm_entries()
    assert np.array_equal(col, g.col) and np.array_equal(val, g.val)
    for f in ("externalCount", "totalSendCount", "indegree", "outdegree"):
        assert getattr(prob, f) == mine[f], f
    for f in ("sources", "recvCounts", "rdispls", "destinations", "sendCounts", "sdispls",
              "elementsToSend", "externalGlobal"):
        assert np.array_equal(prob.array(f), mine[f]), f

    # 2. CG with the PRODUCT's plan, gloo transport, oracle arithmetic
    plan = {f: prob.array(f).copy() for f in ("sources", "recvCounts", "rdispls", "destinations",
                                               "sendCounts", "sdispls", "elementsToSend")}
    nr, nc = prob.nr, prob.nc

    def exchange(p):
        reqs, bufs = [], []
        for i, d in enumerate(plan["destinations"]):
            idx = plan["elementsToSend"][plan["sdispls"][i]:plan["sdispls"][i] + plan["sendCounts"][i]]
            reqs.append(dist.isend(torch.from_numpy(p[idx].copy()), int(d)))
        for i, s in enumerate(plan["sources"]):
            t = torch.zeros(int(plan["recvCounts"][i]), dtype=torch.float64)
            bufs.append((int(plan["rdispls"][i]), t))
            reqs.append(dist.irecv(t, int(s)))
        for q in reqs:
            q.wait()
        for off, t in bufs:
            p[nr + off:nr + off + len(t)] = t.numpy()

    def ddot(a, b):
        t = torch.tensor([po.ddot_seq(a, b)], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0])

    gd = golden[key]
    itermax = gd["itermax"]
    b = g.rhs()
    x, p = np.zeros(nr), np.zeros(nc)
    rr_hist, pap_hist = [], []
    p[:nr] = po.waxpby(1.0, x, 0.0, x)
    exchange(p)
    Ap = g.spmv(p)
    r = po.waxpby(1.0, b, -1.0, Ap)
    rtrans = ddot(r, r)
    rr_hist.append(rtrans)
    normr = np.sqrt(rtrans)
    k = 1
    while k < itermax and normr > 0.0:
        if k == 1:
            p[:nr] = po.waxpby(1.0, r, 0.0, r)
        else:
            old = rtrans
            rtrans = ddot(r, r)
            rr_hist.append(rtrans)
            with np.errstate(all="ignore"):
                beta = float(np.float64(rtrans) / np.float64(old))
            p[:nr] = po.waxpby(1.0, r, beta, p[:nr].copy())
        normr = np.sqrt(rtrans)
        exchange(p)
        Ap = g.spmv(p)
        pap = ddot(p[:nr].copy(), Ap)
        pap_hist.append(pap)
        with np.errstate(all="ignore"):
            alpha = float(np.float64(rtrans) / np.float64(pap))  # 0/0 -> NaN as in C
        x = po.waxpby(1.0, x, alpha, p[:nr].copy())
        r = po.waxpby(1.0, r, -alpha, Ap)
        k += 1
    ref_rr = np.array([float(v) for v in gd["rr"]])
    ref_pap = np.array([float(v) for v in gd["pAp"]])
    assert len(rr_hist) == len(ref_rr), (len(rr_hist), len(ref_rr))
    if size == 2:  # a + b is order independent: bit for bit vs the MPI reference
        assert np.array_equal(np.array(rr_hist), ref_rr)
        assert np.array_equal(np.array(pap_hist), ref_pap)
    else:
        live = ref_rr / ref_rr[0] > 1e-20
        assert (np.abs(np.array(rr_hist) - ref_rr) / ref_rr)[live].max() < 1e-12
    dist.barrier()
    if rank == 0:
        print("DIST_OK", case, size, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
