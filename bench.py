#!/usr/bin/env python3
"""bench.py -- CG iterations/s and SpMV GB/s (fraction of the HBM roofline) on MI355X.

Workload (BASELINE.json configs[2]/[3]): HPCG 27-point stencil, 128^3 rows per GPU,
Sell-C-sigma C=64 sigma=256, fp64 values / u32 indices, synthetic (deterministic
generator, no RNG), one rank per GPU, bricks stacked in z (weak scaling).

A "step" is one CG iteration (loop body of solveCG, src/CGSolver.c:107-129): r.r,
p update, halo exchange, SpMV, p.Ap, x and r updates -- all of it, nothing skipped,
on data already resident in HBM.  Exactly K steps are timed between barrier+sync
pairs; the max over ranks is taken; rank 0 prints ONE JSON line.

  value       = N_gpus * K / seconds   (128^3-brick CG iterations per second summed
                over all GPUs; at N=1 this is plain CG iterations/s)
  roofline    = SpMV kernel: algorithmic bytes / average launch duration
                measured with HIP events on the layer's stream in a second pass of
                the same K iterations
  cpu_baseline= the reference's own solveCG (oracle/_ref, upstream flags + OpenMP)
                timed on this box's host cores on a bounded sample (rank 0, N=1)
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md; ~6.3 TB/s achievable)


def pmc_traffic(workload, kernel):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
    (profiles/*_pmc_traffic.json, newest first) -- PMC counters cannot be collected from inside the
    run; None when no pass exists for this workload / kernel."""
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_traffic.json")), reverse=True) \
            if os.path.isdir(pdir) else []:
        try:
            e = json.load(open(os.path.join(pdir, name))).get(workload, {}).get(kernel)
        except (OSError, ValueError):
            continue
        if e:
            return e["bytes_per_launch"], "profiles/" + name
    return None, None
SEGMENT = 120          # iterations per timed segment (keeps r.r far from underflow)


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
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU
    quota, capped by SB_CPU_CORES (default 16 = a 1-GPU box's CPU share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("SB_CPU_CORES", "16"))))


def cpu_baseline_child(n, iters):
    """Runs in its own process (see cpu_baseline): only oracle/_ref or the oracle is
    loaded there, never the GPU libraries."""
    from oracle import pyoracle as po
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    sample = "HPCG %d^3 CRS, %d CG iterations (solveCG incl. prologue), 1 rank x %d OpenMP threads" % (
        n, iters, threads)
    try:
        if po.ref_available("crs_omp"):
            ref = po.Ref("crs_omp")
            with quiet_stdout():
                ref.setup("generate", n, n, n)
                t0 = time.perf_counter()
                k = ref.L.sbref_solve_cg(iters, 0.0)
                dt = time.perf_counter() - t0
            return {"value": (k - 1) / dt, "unit": "iterations/s", "cores": threads,
                    "kind": "reference", "sample": sample}
    except Exception as e:  # fall through to the port
        sys.stderr.write("cpu_baseline: reference leg failed (%s), using the port\n" % e)
    with quiet_stdout():
        g = po.GMatrix.generate(n, n, n)
        thr = ctypes.c_int(0)
        dt = po.lib().orc_time_cg_iters(g.ptr, iters, ctypes.byref(thr))
    return {"value": iters / dt, "unit": "iterations/s", "cores": thr.value, "kind": "port",
            "sample": sample.replace("solveCG incl. prologue", "loop bodies")}


def cpu_baseline(n, iters):
    """Reference CPU path beside the GPU number: kind 'reference' (its own solveCG,
    -O3 -ffast-math + OpenMP as upstream builds it) or, if oracle/_ref did not travel,
    kind 'port' (the oracle's OpenMP restatement).  Timed in a CHILD process started
    before this process touches the GPU: the reference build carries clang's OpenMP
    runtime, our host library gcc's, and the two must not share a process."""
    import subprocess
    cores = usable_cores()
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(cores))
    env.setdefault("OMP_PROC_BIND", "close")
    env.setdefault("OMP_PLACES", "cores")
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child",
                              "--n", str(n), "--cpu-iters", str(iters)], env=env, check=True,
                             stdout=subprocess.PIPE, timeout=900).stdout.decode()
        return json.loads(out.strip().splitlines()[-1])
    except Exception as e:
        sys.stderr.write("cpu_baseline failed: %s\n" % e)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", "--grid", dest="n", type=int, default=128, help="brick edge per GPU")
    ap.add_argument("--fmt", default="scs", choices=["scs", "crs"])
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--sigma", type=int, default=256)
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--pack-mode", type=int, default=-1,
                    help="SpMV stream: 0 reference layout, 1 compressed, 2 compressed + LDS x-window, "
                         "3 pattern codes / row patterns + LDS x-window (default -1: the library's choice)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="N > 1 data plane: rccl (production) or host (gloo-staged; lets N ranks share one GPU "
                         "to rehearse the multi-rank flow -- its numbers are not a benchmark)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=60)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_child(args.n, args.cpu_iters)), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run "
                             "--nproc-per-node %d\n" % (args.gpus, world, args.gpus))
        if args.gpus > 1:
            sys.exit(2)

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(args.n, args.cpu_iters)  # before the GPU is initialised

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (id broadcast, barriers, max of the timings); the data
        # plane -- halo and dot all-reduce -- is RCCL inside the HIP layer
        with quiet_stdout():  # gloo announces its connections on stdout; rank 0's stdout carries ONE JSON line
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()

    from sparsebench_amd import capi, hostapi
    capi.load()
    ndev = capi.load().sb_device_count()
    L = capi.init(local % ndev if args.transport == "host" and ndev > 0 else local)
    H = hostapi.host()

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

    n = args.n
    with quiet_stdout():
        prob = hostapi.Problem("generate", n, n, n, fmt=args.fmt, Cc=args.C, sigma=args.sigma,
                               rank=rank, size=world)
    cg = hostapi.CG(prob, fused=True, graph=bool(args.graph))
    K, W = args.steps, args.warmup
    # SCS: 0..3; CRS: 0 native kernel, 3 through its private pattern mirror (if the matrix has one)
    mode = prob.use_packed(args.pack_mode) if args.pack_mode >= 0 else prob.pack_info()["mode"]

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

    t_clean, _, _ = timed_pass(False)
    t_ev, spmv_ms, spmv_n = timed_pass(True)
    ref_leg = None
    if mode > 0:  # third pass: the reference-layout kernel (12 B per element), same K iterations
        moved_bytes = prob.stream_bytes()
        pack = prob.pack_info()
        prob.use_packed(0)
        t_ref, ref_ms, ref_n = timed_pass(True)
        prob.use_packed(mode)
        ref_leg = (t_ref, ref_ms, ref_n)
    else:
        moved_bytes, pack = prob.stream_bytes(), {"level": 0, "mode": 0, "lds_window_doubles": 0, "pattern_classes": 0}
    if dist is not None:
        import torch
        tt = torch.tensor([t_clean], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_clean = float(tt[0])

    if rank == 0:
        it_s = K / t_clean
        spmv_us = 1e3 * spmv_ms / max(spmv_n, 1)
        spmv_bytes = prob.spmv_bytes()
        achieved = spmv_bytes / (spmv_us * 1e-6) / 1e9 if spmv_n else 0.0
        nnz_true = prob.nnzTrue
        cg_bytes = 96.0 * prob.nr + spmv_bytes  # reference's unfused op list
        workload = "hpcg_27pt_%d^3_per_gpu_%s_C%d_sigma%d" % (n, args.fmt, args.C, args.sigma)
        native = "spmv_crs_stream" if args.fmt == "crs" else "spmv_scs64"
        kernel = [native, "spmv_scs64_packed", "spmv_scs64_lds", "spmv_scs64_pat"][mode]
        traffic, traffic_src = pmc_traffic(workload, kernel) if world == 1 else (None, None)
        out = {
            "metric": "cg_iterations_per_s",
            "value": world * it_s,
            "unit": "iterations/s (one iteration = one %d^3-brick CG step; summed over GPUs)" % n,
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * t_clean / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "rows_per_gpu": prob.nr, "nnz_per_gpu": nnz_true, "index_type": "u32",
                       "parallelism": "1d_block_row_x%d" % world,
                       "transport": ("none" if world == 1 else "rccl_xgmi" if args.transport == "rccl"
                                     else "host_staged_gloo (rehearsal)"),
                       "halo_exchange": ("none" if world == 1 else "peer_mapped_push_pull" if L.sb_halo_p2p_enabled(prob.halo)
                                         else "rccl_send_recv" if args.transport == "rccl" else "host_staged_gloo"),
                       "dot_allreduce": ("none" if world == 1 else "in_kernel_peer_mapped" if L.sb_comm_p2p_enabled()
                                         else "rccl" if args.transport == "rccl" else "host_staged_gloo"),
                       "fused_dots": True,
                       "hip_graph": bool(args.graph)},
            "global_iterations_per_s": it_s,
            "cg_algorithmic_GBs_per_gpu": cg_bytes * it_s / 1e9,
            "cg_frac_of_hbm_peak": cg_bytes * it_s / 1e9 / HBM_PEAK_GBS,
            "roofline": {"bound": "hbm",
                         "kernel": kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": spmv_bytes,
                         "moved_bytes_per_launch": moved_bytes,
                         "moved_GBs": moved_bytes / (spmv_us * 1e-6) / 1e9 if spmv_n else 0.0,
                         "avg_launch_us": spmv_us, "launches_timed": spmv_n,
                         "ms_per_step_with_events": 1e3 * t_ev / K,
                         "note": ("achieved = bytes of the reference's SCS layout / time; the kernel streams a "
                                  "lossless compressed mirror (moved_bytes), so achieved can exceed the HBM peak"
                                  if mode > 0 else "kernel streams the reference layout")},
            "compression": pack,
            "device": L.sb_device_name().decode(),
        }
        if ref_leg is not None:
            t_ref, ref_ms, ref_n = ref_leg
            ref_us = 1e3 * ref_ms / max(ref_n, 1)
            out["roofline_reference_layout"] = {
                "bound": "hbm", "kernel": native, "achieved": spmv_bytes / (ref_us * 1e-6) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_bytes / (ref_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "avg_launch_us": ref_us, "launches_timed": ref_n,
                "traffic": pmc_traffic(workload, native)[0] if world == 1 else None,
                "cg_iterations_per_s": world * K / t_ref}
        out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    cg.free()
    prob.free()
    if world > 1:
        L.sb_comm_finalize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
