#!/usr/bin/env python3
"""bench.py -- CG iterations/s and SpMV GB/s (fraction of the HBM roofline) on MI355X.

Workloads
  hpcg       (default; BASELINE.json configs[2]/[3]) HPCG 27-point stencil, 128^3 rows per GPU,
             Sell-C-sigma C=64 sigma=256, fp64 values / u32 indices, deterministic generator, one rank
             per GPU, bricks stacked in z (weak scaling).
  irregular  (configs[4]) irregular-nnz stress, CRS vs Sell-C-sigma on one GPU.  SuiteSparse Flan_1565 is
             not available offline; the matrix is the committed stand-in of host/sbh_irregular.c
             (80^3 nodes: 1 536 000 rows, 94 M nonzeros, 3x3-block FE rows of 3..99 entries, 5 % far couplings).

A "step" is one CG iteration (loop body of solveCG, src/CGSolver.c:107-129): r.r, p update, halo
exchange, SpMV, p.Ap, x and r updates -- all of it, nothing skipped, on data already resident in HBM.
Exactly K steps are timed between barrier+sync pairs; the max over ranks is taken; rank 0 prints ONE
JSON line.

  value        = N_gpus * K / seconds (brick CG iterations per second summed over all GPUs; at N=1 this
                 is plain CG iterations/s)
  roofline     = the SpMV kernel that ran in the timed loop: bytes it really moves per launch (matrix
                 stream + x + y, sb_matrix_stream_bytes; for the reference-layout kernels this IS the
                 algorithmic figure of SURVEY 8d) / average launch duration measured with HIP events on
                 the layer's stream in a second pass of the same K iterations.  frac <= 1 by construction.
  roofline_reference_layout = the same for the kernel that streams the reference's own Sell-C-sigma /
                 CRS arrays (12 B per stored element) -- the figure north_star's ">= 60 % of the HBM
                 roofline" is about; algorithmic_speedup = its bytes / the default kernel's bytes.
  cpu_baseline = the reference's own solveCG (oracle/_ref, upstream flags + OpenMP) timed on this box's
                 host cores on a bounded sample (rank 0, N=1)

`python bench.py --gpus N` works as typed: the parent process starts N rank processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, free port) BEFORE anything touches the GPU, relays rank 0's
single JSON line and any non-zero exit code, and never initialises HIP itself.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already exist and each
process is one of them.
"""
import argparse
import contextlib
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md; ~6.3 TB/s achievable)
SEGMENT = 120          # iterations per timed segment (keeps r.r far from underflow)


# ------------------------------------------------------------------------------------------------
# committed PMC passes (rocprofv3 --pmc cannot run inside this process)
# ------------------------------------------------------------------------------------------------
def pmc_traffic(workload, kernel, version):
    """HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, as the guide prescribes) of `kernel` on
    `workload`, from the newest committed profiles/*_pmc_traffic.json whose entry was collected with
    THIS library version.  Returns (bytes, source, note): bytes is None -- never a stale constant --
    when no pass matches, and the note says what is missing."""
    pdir = os.path.join(ROOT, "profiles")
    names = sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_traffic.json")), reverse=True) \
        if os.path.isdir(pdir) else []
    stale = None
    for name in names:
        try:
            doc = json.load(open(os.path.join(pdir, name)))
        except (OSError, ValueError):
            continue
        e = doc.get(workload, {}).get(kernel)
        if not e:
            continue
        if e.get("library_version", doc.get("library_version")) == version:
            return e["bytes_per_launch"], "profiles/" + name, None
        stale = stale or "profiles/%s holds %s/%s for library %r, not %r" % (
            name, workload, kernel, e.get("library_version", doc.get("library_version")), version)
    note = stale or "no committed PMC pass for %s / %s" % (workload, kernel)
    sys.stderr.write("bench: roofline.traffic = null: %s\n" % note)
    return None, None, note


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


# ------------------------------------------------------------------------------------------------
# CPU baseline (child process, never loads the GPU libraries)
# ------------------------------------------------------------------------------------------------
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


def cpu_baseline_child(workload, n, iters):
    """Runs in its own process (see cpu_baseline): only oracle/_ref or the oracle is loaded there,
    never the GPU libraries."""
    import numpy as np
    from oracle import pyoracle as po
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    if workload == "irregular":  # the reference cannot generate it: time the oracle's OpenMP port on the product's matrix
        from sparsebench_amd import hostapi
        with quiet_stdout():
            p = hostapi.Problem("irregular", n, n, n, fmt="crs", upload=False)
            col, val = p.gm_entries()
            g = po.GMatrix.from_csr(p.array("rowPtr").copy(), col, val, nc=p.nc)
            thr = ctypes.c_int(0)
            dt = po.lib().orc_time_cg_iters(g.ptr, iters, ctypes.byref(thr))
        return {"value": iters / dt, "unit": "iterations/s", "cores": thr.value, "kind": "port",
                "sample": "irregular %d^3 nodes CRS, %d CG loop bodies, 1 rank x %d OpenMP threads" % (n, iters, thr.value)}
    sample = "HPCG %d^3 CRS, %d CG loop bodies (difference of two solveCG runs: set-up, prologue and residual check " \
             "cancel), 1 rank x %d OpenMP threads" % (n, iters, threads)
    try:
        if po.ref_available("crs_omp"):
            ref = po.Ref("crs_omp")
            with quiet_stdout():
                ref.setup("generate", n, n, n)
                # two runs: k1 and k2 loop bodies; their difference removes set-up, prologue and check
                t0 = time.perf_counter()
                k1 = ref.L.sbref_solve_cg(max(3, iters // 4), 0.0)
                t1 = time.perf_counter()
                k2 = ref.L.sbref_solve_cg(iters + max(3, iters // 4), 0.0)
                t2 = time.perf_counter()
            dt = (t2 - t1) - (t1 - t0)
            if dt > 0 and k2 > k1:
                return {"value": (k2 - k1) / dt, "unit": "iterations/s", "cores": threads, "kind": "reference",
                        "sample": sample}
    except Exception as e:  # fall through to the port
        sys.stderr.write("cpu_baseline: reference leg failed (%s), using the port\n" % e)
    with quiet_stdout():
        g = po.GMatrix.generate(n, n, n)
        thr = ctypes.c_int(0)
        dt = po.lib().orc_time_cg_iters(g.ptr, iters, ctypes.byref(thr))
    return {"value": iters / dt, "unit": "iterations/s", "cores": thr.value, "kind": "port",
            "sample": sample.replace("the reference's own timeStart/timeStop", "the port")}


def cpu_mpi_leg(n, iters, cores):
    """The reference's hybrid mode (MPI ranks x OpenMP threads), if this box has the MPI launcher the
    reference binary oracle/_ref/sb_ref_mpi_omp was built against.  Returns a dict or None."""
    exe = os.path.join(ROOT, "oracle", "_ref", "sb_ref_mpi_omp")
    mpiexec = os.environ.get("SB_MPIEXEC", "/opt/conda/bin/mpiexec")
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        return None
    best = None
    for ranks in (2, 4, 8):
        if ranks > cores or n % ranks:
            continue
        thr = max(1, cores // ranks)
        # no OMP_PLACES here: every rank would pin its threads to the SAME first cores; hydra spreads the ranks
        env = dict(os.environ, OMP_NUM_THREADS=str(thr), OMP_PROC_BIND="false",
                   PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
        env.pop("OMP_PLACES", None)

        def run(k):
            out = subprocess.run([mpiexec, "-n", str(ranks), "-bind-to", "none", exe, "-x", str(n), "-y", str(n), "-z", str(n // ranks),
                                  "-i", str(k)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
            import re
            m = re.search(r"Solution performed (\d+) iterations and took ([0-9.]+)s", out.stdout.decode())
            return (int(m.group(1)), float(m.group(2))) if m else None
        try:
            a = run(5 * iters + 1)  # (the reference prints its loop time with two decimals: run long enough for that)
        except Exception:
            return best
        if not a or a[1] <= 0:
            continue
        v = (a[0] - 1) / a[1]  # the reference's own loop clock (src/CGSolver.c:106,130), 2-digit resolution
        if best is None or v > best["value"]:
            best = {"value": v, "ranks": ranks, "threads_per_rank": thr, "cores": ranks * thr,
                    "sample": "HPCG %d^3 (z split over ranks) CRS, %d loop bodies, mpiexec -n %d x %d OpenMP threads, "
                              "the reference's own loop clock" % (n, a[0] - 1, ranks, thr)}
    return best


def cpu_baseline(workload, n, iters):
    """Reference CPU path beside the GPU number: kind 'reference' (its own solveCG, -O3 -ffast-math +
    OpenMP as upstream builds it; best of 1 rank x T threads and, where an MPI launcher exists, P ranks
    x T threads) or, if oracle/_ref did not travel, kind 'port' (the oracle's OpenMP restatement).
    Timed in CHILD processes started before this process touches the GPU: the reference build carries
    clang's OpenMP runtime, our host library gcc's, and the two must not share a process."""
    nproc, usable = host_cores()
    cores = max(1, min(usable, int(os.environ.get("SB_CPU_CORES", str(usable)))))
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(cores))
    env.setdefault("OMP_PROC_BIND", "close")
    env.setdefault("OMP_PLACES", "cores")
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--workload", workload,
                              "--n", str(n), "--cpu-iters", str(iters)], env=env, check=True,
                             stdout=subprocess.PIPE, timeout=900).stdout.decode()
        res = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    except Exception as e:
        sys.stderr.write("cpu_baseline failed: %s\n" % e)
        return None
    res["nproc"], res["usable_cores"] = nproc, usable
    res["openmp_only"] = {"value": res["value"], "cores": res["cores"]}
    if workload == "hpcg" and res.get("kind") == "reference" and not os.environ.get("SB_NO_MPI_BASELINE"):
        try:
            mpi = cpu_mpi_leg(n, iters, cores)
        except Exception as e:
            mpi = None
            sys.stderr.write("cpu_baseline: MPI leg failed (%s)\n" % e)
        res["mpi_openmp"] = mpi
        if mpi and mpi["value"] > res["value"]:
            res.update(value=mpi["value"], cores=mpi["cores"], sample=mpi["sample"])
    return res


# ------------------------------------------------------------------------------------------------
# N > 1: the parent starts the ranks itself
# ------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n_gpus, argv):
    """One child per rank; this process never touches the GPU (no HIP call, no exec of a process that
    has).  Rank 0's stdout is captured and its single JSON line relayed; the others' stdout goes to
    stderr.  Exit code = the first non-zero child exit code."""
    port = free_port()
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SB_BENCH_RANK_PROCESS="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = b""
    rc = 0
    try:
        out0, _ = procs[0].communicate()
        for p in procs:
            code = p.wait()
            if code != 0 and rc == 0:
                rc = code
    finally:
        for p in procs:  # a rank that died leaves the others waiting in a collective: end exactly those we started
            if p.poll() is None:
                p.kill()
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc = 3
        sys.stderr.write("bench: rank 0 printed no JSON line\n")
    return rc


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def kernel_name(fmt, mode):
    native = "spmv_crs_stream" if fmt == "crs" else "spmv_scs64"
    return {0: native, 1: "spmv_scs64_packed", 2: "spmv_scs64_lds", 3: "spmv_scs64_pat", 5: "spmv_scs64_pat_masked"}[mode]


def roofline_block(kernel, moved, alg, us, launches, traffic, traffic_src, traffic_note):
    gbs = moved / (us * 1e-6) / 1e9 if launches else 0.0
    blk = {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
           "bytes_per_launch": moved, "algorithmic_bytes_per_launch": alg,
           "avg_launch_us": us, "launches_timed": launches}
    if traffic:
        blk["traffic_over_bytes"] = traffic / moved
        if launches:  # the same fraction on the bytes the PMC counters saw (gathers that miss the caches included)
            blk["frac_on_traffic"] = traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
    if traffic_note:
        blk["traffic_note"] = traffic_note
    return blk


def run_rank(args):
    import numpy as np  # noqa: F401
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    irregular = args.workload == "irregular"
    n = args.n if args.n > 0 else (80 if irregular else 128)

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(args.workload, n, args.cpu_iters)  # before the GPU is initialised

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (id broadcast, barriers, max of the timings); the data plane -- halo and dot
        # all-reduce -- lives inside the HIP layer
        with quiet_stdout():  # gloo announces its connections on stdout; rank 0's stdout carries ONE JSON line
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()

    from sparsebench_amd import capi, hostapi
    capi.load()
    ndev = capi.load().sb_device_count()
    L = capi.init(local % ndev if args.transport == "host" and ndev > 0 else local)
    if world > max(ndev, 1):  # ranks share GPUs (rehearsal): the one-launch vector phase needs a GPU to itself
        os.environ.setdefault("SB_SHARED_GPU", "1")
    H = hostapi.host()
    version = L.sb_version().decode()

    keep = None
    if world > 1 and args.transport == "host":
        from sparsebench_amd import gloo_transport
        keep = gloo_transport.attach(L, H, dist, rank, world)  # noqa: F841
    elif world > 1:
        import torch
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            raw = (ctypes.c_ubyte * 128)()
            L.sb_comm_unique_id(raw)
            idbuf = torch.tensor(list(raw), dtype=torch.uint8)
        dist.broadcast(idbuf, 0)
        raw = (ctypes.c_ubyte * 128)(*idbuf.tolist())
        L.sb_comm_init(rank, world, raw)
        H.commSetExchange(H.sbh_exchange_rccl())

    def barrier():
        L.sb_sync()
        if dist is not None:
            dist.barrier()
        L.sb_sync()

    K, W = args.steps, args.warmup
    vphase, launches = 0, 5

    def measure(prob, modes):
        """timed passes on one resident matrix.  modes: kernel modes to time with per-launch events; the
        first is the one `value` is quoted on (clean pass without events)."""
        cg = hostapi.CG(prob, fused=args.fused, graph=bool(args.graph))
        nonlocal vphase, launches
        vphase, launches = cg.vector_phase(), cg.launches_per_body()

        def timed_pass(with_spmv_events):
            """exactly K loop bodies, in segments restarted from x0 = 0 outside the clock"""
            total, left, spmv_ms, spmv_n = 0.0, K, 0.0, 0
            while left > 0:
                seg = min(left, SEGMENT)
                cg.spmv_timing(False)
                cg.start(itermax=W + 2 + seg, eps=0.0)  # prologue
                cg.run_iters(W + 1)                     # warm-up bodies, untimed
                before = cg.counters()
                cg.spmv_timing(with_spmv_events)
                barrier()
                t0 = time.perf_counter()
                cg.run_iters(seg)
                L.sb_sync()  # this rank's K steps are complete on its GPU ...
                dt = time.perf_counter() - t0
                barrier()    # ... and nobody moves on before all are (the max over ranks is taken below;
                #                 the gloo TCP barrier itself is control plane, not part of a CG step)
                after = cg.counters()
                cg.finish()
                if after["stop"] and after["iters"] != W + 1 + seg:
                    raise RuntimeError("bench: the loop exited early: %r" % after)
                if after["n_pAp"] - before["n_pAp"] != seg or after["iters"] != W + 1 + seg:
                    raise RuntimeError("bench: the timed iterations did not all execute: %r -> %r" % (before, after))
                if with_spmv_events:
                    ms, cnt = cg.spmv_ms()
                    spmv_ms += ms
                    spmv_n += cnt
                total += dt
                left -= seg
            return total, spmv_ms, spmv_n

        res = {}
        for i, mode in enumerate(modes):
            got = prob.use_packed(mode)
            if got != mode:
                continue
            t_clean = timed_pass(False)[0] if i == 0 or args.all_clean else None
            t_ev, ms, cnt = timed_pass(True)
            if t_clean is not None and dist is not None:
                import torch
                tt = torch.tensor([t_clean], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t_clean = float(tt[0])
            res[mode] = {"t_clean": t_clean, "t_ev": t_ev, "spmv_us": 1e3 * ms / max(cnt, 1), "launches": cnt,
                         "moved": prob.stream_bytes(), "alg": prob.spmv_bytes()}
        prob.use_packed(modes[0])
        cg.free()
        return res

    def vector_bytes(nr):
        """bytes the fused loop's vector kernels move per iteration.  Separate launches: p update (+ the x update
        owed by the previous body) 40 B/row, r update + r.r partials 24 B/row.  One-launch vector phase: r, Ap, p, x
        read and r, p, x written once: 56 B/row.  Plus the partials written and read back."""
        return (56.0 if vphase else 64.0) * nr + 2 * 8.0 * (nr / 64.0)

    out = None
    if not irregular:
        with quiet_stdout():
            prob = hostapi.Problem("generate", n, n, n, fmt=args.fmt, Cc=args.C, sigma=args.sigma, rank=rank, size=world)
        default = prob.use_packed(args.pack_mode) if args.pack_mode >= 0 else prob.pack_info()["mode"]
        modes = [default] + ([0] if default != 0 else [])
        res = measure(prob, modes)
        rank_modes = [default]
        if dist is not None:  # which SpMV kernel every rank ran (rank-local matrices differ: halo above / below / both)
            import torch
            mine = torch.tensor([default], dtype=torch.int32)
            got = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
            dist.all_gather(got, mine)
            rank_modes = [int(t[0]) for t in got]
        if rank == 0:
            d = res[default]
            it_s = K / d["t_clean"]
            workload = "hpcg_27pt_%d^3_per_gpu_%s_C%d_sigma%d" % (n, args.fmt, args.C, args.sigma)
            kern = kernel_name(args.fmt, default)
            tr = pmc_traffic(workload, kern, version) if world == 1 else (None, None, "N > 1")
            cg_moved = d["moved"] + vector_bytes(prob.nr)
            cg_alg = 96.0 * prob.nr + d["alg"]  # reference's unfused op list on its own layout
            out = {
                "metric": "cg_iterations_per_s",
                "value": world * it_s,
                "unit": "iterations/s (one iteration = one %d^3-brick CG step; summed over GPUs)" % n,
                "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": 1e3 * d["t_clean"] / K,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": workload,
                           "rows_per_gpu": prob.nr, "nnz_per_gpu": prob.nnzTrue, "index_type": "u32",
                           "parallelism": "1d_block_row_x%d" % world,
                           "transport": ("none" if world == 1 else "rccl_xgmi" if args.transport == "rccl"
                                         else "host_staged_gloo (rehearsal)"),
                           "halo_exchange": ("none" if world == 1 else "peer_mapped_push_pull" if L.sb_halo_p2p_enabled(prob.halo)
                                             else "rccl_send_recv" if args.transport == "rccl" else "host_staged_gloo"),
                           "dot_allreduce": ("none" if world == 1 else "in_kernel_peer_mapped" if L.sb_comm_p2p_enabled()
                                             else "rccl" if args.transport == "rccl" else "host_staged_gloo"),
                           "dot_allreduce_reason": (L.sb_comm_p2p_reason().decode() if world > 1 else None),
                           "halo_exchange_reason": (L.sb_halo_p2p_reason(prob.halo).decode() if world > 1 else None),
                           "spmv_x_staging": ("lds_window" if default >= 2 else "l1_l2_gather (LDS staging measured neutral at 12 B/element)"),
                           "fused_dots": True, "vector_phase_one_launch": bool(vphase), "launches_per_iteration": launches, "spmv_kernel_mode_by_rank": rank_modes,
                           "hip_graph": bool(args.graph), "library": version},
                "global_iterations_per_s": it_s,
                "roofline": roofline_block(kern, d["moved"], d["alg"], d["spmv_us"], d["launches"], *tr),
                "algorithmic_speedup": d["alg"] / d["moved"],
                "cg_moved_bytes_per_iteration": cg_moved,
                "cg_moved_GBs_per_gpu": cg_moved * it_s / 1e9,
                "cg_frac_of_hbm_peak": cg_moved * it_s / 1e9 / HBM_PEAK_GBS,
                "cg_reference_oplist_bytes_per_iteration": cg_alg,
                "ms_per_step_with_events": 1e3 * d["t_ev"] / K,
                "compression": prob.pack_info(),
                "device": L.sb_device_name().decode(),
            }
            out["roofline"]["note"] = (
                "bytes = what this kernel streams (lossless compressed mirror, %.1f MB instead of the reference layout's "
                "%.1f MB): a real HBM fraction; the reference-layout kernel is roofline_reference_layout" % (
                    d["moved"] / 1e6, d["alg"] / 1e6) if default > 0 else "kernel streams the reference layout: bytes = SURVEY 8d")
            if default != 0 and 0 in res:
                r0 = res[0]
                k0 = kernel_name(args.fmt, 0)
                tr0 = pmc_traffic(workload, k0, version) if world == 1 else (None, None, "N > 1")
                out["roofline_reference_layout"] = roofline_block(k0, r0["moved"], r0["alg"], r0["spmv_us"], r0["launches"], *tr0)
                out["roofline_reference_layout"]["cg_iterations_per_s_with_events"] = world * K / r0["t_ev"]
        prob.free()
    else:
        if world != 1:
            raise SystemExit("bench: --workload irregular is a one-GPU workload (configs[4])")
        formats = {}
        best = None
        specs = [("crs", 1)] + [("scs", s) for s in args.irr_sigmas]
        for fmt, sigma in specs:
            with quiet_stdout():
                prob = hostapi.Problem("irregular", n, n, n, fmt=fmt, Cc=64, sigma=sigma)
            default = prob.pack_info()["mode"]
            res = measure(prob, [default])
            d = res[default]
            name = "crs" if fmt == "crs" else "scs_C64_sigma%d" % sigma
            workload = "irregular_fe_%d^3_nodes_%s" % (n, name)
            kern = kernel_name(fmt, default)
            tr = pmc_traffic(workload, kern, version)
            # (the native CRS kernel has no fused p.Ap: the loop adds a dot pass over p and Ap)
            cg_moved = d["moved"] + vector_bytes(prob.nr) + (16.0 * prob.nr if fmt == "crs" and default == 0 else 0.0)
            formats[name] = {
                "cg_iterations_per_s": K / d["t_clean"], "ms_per_step": 1e3 * d["t_clean"] / K,
                "fill": (prob.nnzTrue / prob.nElems) if fmt == "scs" else 1.0,
                "roofline": roofline_block(kern, d["moved"], d["alg"], d["spmv_us"], d["launches"], *tr),
                "spmv_useful_GBs": (12.0 * prob.nnzTrue + 16.0 * prob.nr) / (d["spmv_us"] * 1e-6) / 1e9,
                "cg_frac_of_hbm_peak": cg_moved * (K / d["t_clean"]) / 1e9 / HBM_PEAK_GBS}
            if best is None or formats[name]["cg_iterations_per_s"] > formats[best]["cg_iterations_per_s"]:
                best = name
                meta = {"rows": prob.nr, "nnz": prob.nnzTrue}
            prob.free()
        b = formats[best]
        out = {
            "metric": "cg_iterations_per_s", "value": b["cg_iterations_per_s"],
            "unit": "iterations/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": b["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "irregular_fe_%d^3_nodes (SuiteSparse Flan_1565 not available offline; committed stand-in "
                                   "host/sbh_irregular.c), best format: %s" % (n, best),
                       "rows_per_gpu": meta["rows"], "nnz_per_gpu": meta["nnz"], "index_type": "u32",
                       "parallelism": "1d_block_row_x1", "library": version},
            "roofline": b["roofline"], "formats": formats, "device": L.sb_device_name().decode(),
        }

    if rank == 0 and out is not None:
        out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        L.sb_comm_finalize()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="hpcg", choices=["hpcg", "irregular"])
    ap.add_argument("--n", "--grid", dest="n", type=int, default=0,
                    help="hpcg: brick edge per GPU (default 128); irregular: nodes per mesh edge (default 80)")
    ap.add_argument("--fmt", default="scs", choices=["scs", "crs"])
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--sigma", type=int, default=256)
    ap.add_argument("--irr-sigmas", type=lambda s: [int(v) for v in s.split(",")], default=[1, 4096],
                    help="irregular: sorting scopes of the Sell-C-sigma legs")
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--fused", type=int, default=1,
                    help="sb_cg_set_fused level: 1 five launches per loop body (default), 2 / 3 / 4 the measured alternatives")
    ap.add_argument("--pack-mode", type=int, default=-1,
                    help="SpMV stream: 0 reference layout, 1 compressed, 2 compressed + LDS x-window, "
                         "3 pattern codes / row patterns + LDS x-window (default -1: the library's choice)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="N > 1 data plane: rccl (production) or host (gloo-staged; lets N ranks share one GPU "
                         "to rehearse the multi-rank flow -- its numbers are not a benchmark)")
    ap.add_argument("--all-clean", action="store_true", help="also time the secondary kernel modes without events")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=60)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        n = args.n if args.n > 0 else (80 if args.workload == "irregular" else 128)
        print(json.dumps(cpu_baseline_child(args.workload, n, args.cpu_iters)), flush=True)
        return 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        if os.environ.get("SB_BENCH_RANK_PROCESS"):
            sys.stderr.write("bench: rank process with WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
            return 2
        return spawn_ranks(args.gpus, sys.argv[1:])  # before any HIP call; this process stays off the GPU
    if args.gpus == 1 and world > 1:
        sys.stderr.write("bench: --gpus 1 but WORLD_SIZE=%d\n" % world)
        return 2
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
