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
from sparsebench_amd import capi, hostapi  # noqa: E402

vp = C.c_void_p


def main():
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    fmt, Cc, sigma, n, itermax = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    L = capi.init(0)  # every rank on the one GPU
    H = hostapi.host()

    # ---- setup exchange of commPartition over gloo (as tests/dist_worker.py) -------------
    def allgather(ctx, mine, cnt, out):
        t = torch.tensor([mine[i] for i in range(cnt)], dtype=torch.int32)
        outs = [torch.zeros(cnt, dtype=torch.int32) for _ in range(size)]
        dist.all_gather(outs, t)
        for i, v in enumerate(torch.cat(outs).tolist()):
            out[i] = v

    def alltoallv(ctx, sbuf, scnt, sdsp, rbuf, rcnt, rdsp):
        reqs, recv = [], {}
        for r in range(size):
            if r == rank:
                continue
            if scnt[r]:
                reqs.append(dist.isend(torch.tensor([sbuf[sdsp[r] + i] for i in range(scnt[r])], dtype=torch.int32), r))
            if rcnt[r]:
                recv[r] = torch.zeros(rcnt[r], dtype=torch.int32)
                reqs.append(dist.irecv(recv[r], r))
        for q in reqs:
            q.wait()
        for r, t in recv.items():
            for i, v in enumerate(t.tolist()):
                rbuf[rdsp[r] + i] = v

    cb1, cb2 = hostapi.ALLGATHER_FN(allgather), hostapi.ALLTOALLV_FN(alltoallv)
    xchg = hostapi.ExchangeS(None, cb1, cb2)
    H.commSetExchange(C.byref(xchg))

    # ---- run-time transport: device buffers staged through the host, gloo in between -----
    def d2h(ptr, count):
        a = np.empty(count, dtype=np.float64)
        if count:
            L.sb_d2h(a.ctypes.data_as(vp), ptr, count * 8)
        return a

    def allreduce(ctx, v_dev, op):
        mine = torch.from_numpy(d2h(v_dev, 1))
        outs = [torch.zeros(1, dtype=torch.float64) for _ in range(size)]
        dist.all_gather(outs, mine)
        vals = [float(t[0]) for t in outs]
        if op == 0:
            res = max(vals)
        else:  # pairwise tree == recursive doubling of the MPI reference run
            while len(vals) > 1:
                nxt = [vals[i] + vals[i + 1] for i in range(0, len(vals) - 1, 2)]
                if len(vals) & 1:
                    nxt.append(vals[-1])
                vals = nxt
            res = vals[0]
        out = np.array([res])
        L.sb_h2d(v_dev, out.ctypes.data_as(vp), 8)

    def exchange(ctx, send_dev, outdeg, dests, scnt, sdsp, recv_dev, indeg, srcs, rcnt, rdsp):
        total = sum(scnt[i] for i in range(outdeg))
        sbuf = d2h(send_dev, total)
        reqs, bufs = [], []
        for i in range(outdeg):
            reqs.append(dist.isend(torch.from_numpy(sbuf[sdsp[i]:sdsp[i] + scnt[i]].copy()), dests[i]))
        for j in range(indeg):
            t = torch.zeros(rcnt[j], dtype=torch.float64)
            bufs.append((rdsp[j], t))
            reqs.append(dist.irecv(t, srcs[j]))
        for q in reqs:
            q.wait()
        for off, t in bufs:
            a = t.numpy()
            L.sb_h2d(recv_dev + off * 8, a.ctypes.data_as(vp), len(a) * 8)

    cb3, cb4 = capi.ALLREDUCE_FN(allreduce), capi.EXCHANGE_FN(exchange)
    tr = capi.TransportS(None, cb3, cb4)
    L.sb_comm_init_transport(rank, size, C.byref(tr))

    prob = hostapi.Problem("generate", n, n, n, fmt=fmt, Cc=Cc, sigma=sigma, rank=rank, size=size)
    locs = [po.GMatrix.generate(n, n, n, r, size) for r in range(size)]
    plans = po.Plans(locs)
    o = po.cg(locs, plans, itermax=itermax, fmt=fmt, Cc=Cc, sigma=sigma, dot="tree", rank_sum="tree", want_x=True)
    results = {}
    for mode in ((3, 2, 1, 0) if fmt == "scs" and Cc == 64 else (0,)):
        if fmt == "scs":
            prob.use_packed(mode)
        for fused in (True, False):
            cg = hostapi.CG(prob, fused=fused)
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
            results[(mode, fused)] = rr
    key = "hpcg%d_x%d" % (n, size)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "cg_hist_mpi.json")))
    if key in golden and golden[key]["itermax"] == itermax:
        ref = np.array([float(v) for v in golden[key]["rr"]])
        rr = results[(0 if fmt == "crs" else 2 if (2, True) in results else 0, True)]
        live = ref / ref[0] >= 1e-20
        assert (np.abs(rr - ref) / ref)[live].max() <= 1e-12  # north_star tolerance vs the MPI reference
    dist.barrier()
    prob.free()
    L.sb_comm_finalize()
    if rank == 0:
        print("GPU_MULTIRANK_OK", fmt, Cc, sigma, n, size, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
