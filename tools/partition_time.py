#!/usr/bin/env python3
"""partition_time.py n -- wall time of generate + commPartition + convertMatrix for the irregular stand-in (configs[4]:
n = 80 is Flan_1565's class, 1.5 M rows / 94 M nonzeros) on P ranks, CPU only (setup exchange over gloo, no upload):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 tools/partition_time.py 80
Round 3, build container (8 cores, OMP_NUM_THREADS=2): P = 3: 3.8-3.9 s per rank (309-323 k external columns each, every rank a
neighbour of every other); P = 8: 2.8-3.0 s (176-193 k external columns, indegree 7).  The reference's partitioner is a BST plus an
O(externals^2) grouping (src/comm.c:66-78): ~1e11 steps at these counts."""
import os, sys, time
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebench_amd import hostapi, gloo_transport
import ctypes as C
dist.init_process_group("gloo")
rank, size = dist.get_rank(), dist.get_world_size()
H = hostapi.host()
# setup exchange only (no device): reuse the callbacks of gloo_transport without the device part
import torch
def allgather(ctx, mine, cnt, out):
    t = torch.tensor([mine[i] for i in range(cnt)], dtype=torch.int32)
    outs = [torch.zeros(cnt, dtype=torch.int32) for _ in range(size)]
    dist.all_gather(outs, t)
    for i, v in enumerate(torch.cat(outs).tolist()):
        out[i] = v
def alltoallv(ctx, sbuf, scnt, sdsp, rbuf, rcnt, rdsp):
    import numpy as np
    reqs, recv = [], {}
    for r in range(size):
        if r == rank: continue
        if scnt[r]:
            a = np.ctypeslib.as_array(C.cast(C.addressof(sbuf.contents) + 4 * sdsp[r], C.POINTER(C.c_int32)), shape=(scnt[r],)).copy()
            reqs.append(dist.isend(torch.from_numpy(a), r))
        if rcnt[r]:
            recv[r] = torch.zeros(rcnt[r], dtype=torch.int32)
            reqs.append(dist.irecv(recv[r], r))
    for q in reqs: q.wait()
    for r, t in recv.items():
        C.memmove(C.addressof(rbuf.contents) + 4 * rdsp[r], t.numpy().ctypes.data, 4 * rcnt[r])
cb1, cb2 = hostapi.ALLGATHER_FN(allgather), hostapi.ALLTOALLV_FN(alltoallv)
x = hostapi.ExchangeS(None, cb1, cb2)
H.commSetExchange(C.byref(x))
n = int(sys.argv[1])
t0 = time.time()
p = hostapi.Problem("irregular", n, n, n, fmt="crs", rank=rank, size=size, upload=False)
t1 = time.time()
print("rank %d/%d: irregular %d^3: %d rows, %d nnz, %d external columns, indegree %d: generate + commPartition + convertMatrix %.2f s (setup_seconds %.2f)" % (
    rank, size, n, p.nr, p.nnzTrue, p.externalCount, p.indegree, t1 - t0, p.setup_seconds), flush=True)
dist.barrier()
