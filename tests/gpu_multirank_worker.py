"""Worker of tests/test_gpu_multirank.py: P processes share GPU 0 and run the REAL
multi-rank device path of the product (partition -> upload -> halo plan in HBM -> CG with
pack kernel, halo exchange into the tail of p, per-dot all-reduce, device-side loop test),
with torch.distributed/gloo as a host-mediated transport (include/sbhip.h: sb_transport)
in place of RCCL.  History must equal (a) the oracle's P-rank run with the same dot and
rank-sum order, bit for bit, and (b) the MPI reference within the documented tolerance
(bit for bit when the format/sigma leaves the dot order sequential-compatible is not
expected: the GPU dot is the fixed tree order)."""
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
from sparsebench_amd import capi, gloo_transport, hostapi  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from irregular_locs import irregular_locs  # noqa: E402

vp = C.c_void_p


def main():
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    fmt, Cc, sigma, n, itermax = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    workload = sys.argv[6] if len(sys.argv) > 6 else "generate"  # or "irregular": the configs[4] stand-in, n^3 nodes
    os.environ.setdefault("SB_SHARED_GPU", "1")  # every rank on the one GPU: no placement search (its timings would be each other's noise)
    L = capi.init(0)  # every rank on the one GPU
    H = hostapi.host()

    # setup exchange of commPartition + run-time transport (device buffers staged through
    # the host, gloo in between): sparsebench_amd/gloo_transport.py
    keep = gloo_transport.attach(L, H, dist, rank, size)  # noqa: F841  (ctypes callbacks must stay alive)

    prob = hostapi.Problem(workload, n, n, n, fmt=fmt, Cc=Cc, sigma=sigma, rank=rank, size=size)
    # (irregular: far couplings make every rank a neighbour of every other; general-matrix kernels, no pattern levels)
    locs = irregular_locs(n, size) if workload == "irregular" else [po.GMatrix.generate(n, n, n, r, size) for r in range(size)]
    plans = po.Plans(locs)
    o = po.cg(locs, plans, itermax=itermax, fmt=fmt, Cc=Cc, sigma=sigma, dot="tree", rank_sum="tree", want_x=True)
    results = {}
    # every kernel the matrix has: SCS C=64 levels 0..3; CRS native (0) and through its pattern mirror (3)
    vphase_seen = 0
    # (every rank walks the SAME list: what a mode is clamped to may differ from rank to rank, the number of solves must not)
    lab = bool(L.sb_lab_build())  # (the product: kernel modes 5 / 0 and fused 1 / 0; lab builds walk the alternatives too)
    for mode in (((5, 3, 2, 1, 0) if lab else (5, 0)) if fmt == "scs" and Cc == 64 else ((5, 3, 0) if lab else (5, 0)) if fmt == "crs" else (0,)):
        prob.use_packed(mode)  # clamped to what the matrix has
        # 2: the vector phase as one launch (with the in-kernel all-reduce only; the test caps its grid so that the
        # grids of all ranks on the one GPU are resident together), 1: five launches per body, 0: reference op list
        # (3, scalar steps inside their consumers, is a one-rank mode -- DESIGN 4.4 says why -- and behaves as 1 here)
        for fused in ((2, 1, 0) if lab else (1, 0)):
            cg = hostapi.CG(prob, fused=fused)
            vphase_seen += cg.vector_phase() > 0
            if os.environ.get("SB_TEST_VERBOSE"):
                print("rank %d: mode %d fused %d vector phase %d" % (rank, mode, fused, cg.vector_phase()), flush=True)
            k = cg.solve(itermax, 0.0)
            rr, pap = cg.history()
            x = cg.solution()
            err = cg.check_residual()
            cg.free()
            assert k == o["k"], (k, o["k"])
            assert np.array_equal(rr, o["rr"]), ("rr", mode, fused, rank)
            assert np.array_equal(pap, o["pAp"]), ("pAp", mode, fused, rank)
            assert np.array_equal(x, o["x"][rank]), ("x", mode, fused, rank)
            assert err == o["max_err"]
            results[(mode, bool(fused))] = rr
    key = "hpcg%d_x%d" % (n, size)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "cg_hist_mpi.json")))
    if workload == "generate" and key in golden and golden[key]["itermax"] == itermax:
        ref = np.array([float(v) for v in golden[key]["rr"]])
        rr = results[(0 if fmt == "crs" else 2 if (2, True) in results else 0, True)]
        live = ref / ref[0] >= 1e-20
        assert (np.abs(rr - ref) / ref)[live].max() <= 1e-12  # north_star tolerance vs the MPI reference
    dist.barrier()
    print("VPHASE_RUNS %d" % vphase_seen, flush=True)
    mch = C.c_uint32(0)
    L.sb_matrix_row_programs(prob.matrix, C.byref(mch))
    nchunks = (prob.nr + 63) // 64
    print("ROW_PROGRAMS rank %d %d of %d" % (rank, mch.value, nchunks if fmt == "scs" and Cc == 64 else 0), flush=True)
    # the data plane every rank REALLY used (both set-ups are collective decisions: all ranks agree)
    p2p, halo_p2p = L.sb_comm_p2p_enabled(), L.sb_halo_p2p_enabled(prob.halo)
    flags = torch.tensor([p2p, halo_p2p], dtype=torch.int32)
    lo, hi = flags.clone(), flags.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi), "ranks disagree on the data plane: %r vs %r" % (lo.tolist(), hi.tolist())
    why, why_halo = L.sb_comm_p2p_reason().decode(), L.sb_halo_p2p_reason(prob.halo).decode()
    prob_indegree = prob.indegree
    prob.free()
    L.sb_comm_finalize()
    if rank == 0:
        print("P2P_ENABLED", p2p, flush=True)
        print("HALO_P2P_ENABLED", halo_p2p, flush=True)
        print("P2P_REASON", why, flush=True)
        print("HALO_P2P_REASON", why_halo, flush=True)
        print("GPU_MULTIRANK_OK", fmt, Cc, sigma, n, size, flush=True)
        print("INDEGREE", prob_indegree, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
