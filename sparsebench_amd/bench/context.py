"""What every leg of a rank's bench run shares: rank / world, the C-ABI library handles, the control plane (gloo, N > 1 only:
id broadcast, barriers, max of the timings -- the data plane lives inside the HIP layer) and the timing parameters."""
import contextlib
import ctypes
import os
import sys


@contextlib.contextmanager
def quiet_stdout():
    """C code under us prints (generator banner, reference solver): keep stdout clean"""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        try:
            ctypes.CDLL(None).fflush(None)  # the C side's buffered lines go to /dev/null too, not out at exit
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)


def host_cores():
    """(nproc, usable): cores of the machine, and those this process may really use (affinity mask
    capped by the cgroup CPU quota)"""
    nproc = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = nproc
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return nproc, max(1, n)


class RankContext:
    """One rank of the run.  Creating it initialises the control plane, the device and (N > 1) the data plane's communicator."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.K, self.W = args.steps, args.warmup
        self.repeats = 1 if self.K >= 100 else 9  # a 1 ms window moves by a few per cent from run to run: median of 9
        self.supervised = bool(os.environ.get("SB_BENCH_RANK_PROCESS"))
        self.dist = None
        rank, world = self.rank, self.world
        if world > 1:
            import torch  # noqa: F401
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with quiet_stdout():  # gloo announces its connections on stdout; rank 0's stdout carries ONE JSON line
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.barrier()
            # set-up (generator, partitioner, layout) is OpenMP-parallel on the host: share the cores between the ranks
            os.environ.setdefault("OMP_NUM_THREADS", str(max(1, host_cores()[1] // world)))
        if os.environ.get("SB_BENCH_TEST_DIE_RANK") == str(rank):  # test hook: a rank that dies before the first collective
            sys.stderr.write("bench: rank %d: SB_BENCH_TEST_DIE_RANK is set, exiting with code 7 (test hook)\n" % rank)
            os._exit(7)

        from sparsebench_amd import capi, hostapi
        capi.load()
        ndev = capi.load().sb_device_count()
        self.device = self.local % ndev if args.transport == "host" and ndev > 0 else self.local
        self.L = L = capi.init(self.device)
        if world > max(ndev, 1):  # ranks share GPUs (rehearsal): the one-launch vector phase needs a GPU to itself
            os.environ.setdefault("SB_SHARED_GPU", "1")
        self.H = H = hostapi.host()
        self.version = L.sb_version().decode()
        self._keep = None
        if world > 1 and args.transport == "host":
            from sparsebench_amd import gloo_transport
            self._keep = gloo_transport.attach(L, H, self.dist, rank, world)
        elif world > 1:
            import torch
            idbuf = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (ctypes.c_ubyte * 128)()
                L.sb_comm_unique_id(raw)
                idbuf = torch.tensor(list(raw), dtype=torch.uint8)
            self.dist.broadcast(idbuf, 0)
            raw = (ctypes.c_ubyte * 128)(*idbuf.tolist())
            L.sb_comm_init(rank, world, raw)
            H.commSetExchange(H.sbh_exchange_rccl())

    def barrier(self):
        self.L.sb_sync()
        if self.dist is not None:
            self.dist.barrier()
        self.L.sb_sync()

    def gather(self, obj):
        """every rank's `obj`, in rank order"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def rank_max(self, v):
        if self.dist is None:
            return float(v)
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def new_cg(self, prob, graph=None):
        from sparsebench_amd import hostapi
        a = self.args
        return hostapi.CG(prob, fused=a.fused, graph=bool(a.graph) if graph is None else graph, fuse_p=a.fuse_p,
                          fuse_alpha=a.fuse_alpha, fuse_beta=a.fuse_beta)

    def finalize(self):
        if self.world > 1:
            self.L.sb_comm_finalize()
            self.dist.destroy_process_group()
