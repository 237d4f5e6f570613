#!/usr/bin/env python3
"""bench.py -- CG iterations/s and SpMV GB/s (fraction of the HBM roofline) on MI355X.

Entry point and CPU-baseline legs; the rest lives in sparsebench_amd/bench/ (context: rank set-up and control plane, preflight:
known answers before anything is timed, timing: the timed passes, line: roofline block and byte counts, hpcg / irregular: the two
workloads' flows and lines, supervisor: the N > 1 process supervisor).

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

  value        = N_gpus * K / seconds of the loop whose SpMV STREAMS THE REFERENCE'S OWN ARRAYS (spmv_scs64 / spmv_crs_split,
                 12 B per stored element): the loop SURVEY 8d's roofline is defined on (brick CG iterations per second summed
                 over all GPUs; at N=1 plain CG iterations/s).  ms_per_step, phases_us, sustained describe the same loop.
  roofline     = that SpMV kernel: SURVEY 8d's algorithmic bytes per launch (sb_matrix_spmv_bytes, the true-nnz formula) /
                 average launch duration measured with HIP events on the layer's stream in a second pass of the same K
                 iterations; traffic = HBM bytes per launch from the committed PMC passes (profiles/*_pmc_traffic.json).
  cg_frac_of_roofline = value per GPU x the reference's unfused op-list bytes per iteration (96 B/row + the SpMV's) / 8 TB/s.
  structure_exploiting = the same loop with the SpMV on the lossless compressed mirror (masked row programs, p update inside
                 the SpMV launch): its rate, the bytes it moves, its fraction of the HBM peak on MOVED bytes,
                 algorithmic_speedup, sustained.  A real, bit-identical solver rate; not a roofline figure (its working set
                 lives in the Infinity Cache).
  cpu_baseline = the reference's own solveCG (oracle/_ref, upstream flags + OpenMP) timed on this box's
                 host cores on a bounded sample (rank 0, N=1)
  preflight    = before anything is timed, every rank solves 20 CG iterations on a 32^3-per-rank problem and on
                 the bench's own bricks, with every SpMV kernel that will be timed, and checks the history against
                 closed-form known answers (r.r of the prologue, p.Ap of the first body: exact integers at any size and
                 rank count, sparsebench_amd/knownanswers.py), against the committed oracle histories in the GPU's dot order
                 (tests/golden/cg_hist_tree.json: bit for bit), and against every other rank's history (identical
                 bits).  Any mismatch: no rate is printed from that data plane (one plane only: exit code 4).  At N > 1
                 this runs on BOTH data planes, each before it is timed.
                 (irregular workload: the stand-in at 24^3 nodes against the history the reference itself produced on it,
                 every format within 1e-12, CRS and Sell-64-1 identical bits.)
  rccl_only    = (N > 1) the same K steps timed with the peer-mapped paths switched off
                 (sb_comm_data_plane(0): RCCL all-reduce + send/recv), so one invocation yields both curves.
  roofline.device_stream_read_GBs / achieved_over_device_stream_read = (N = 1, informational) a plain streaming read of a fresh
                 1 GiB buffer measured in the same process after the timed passes, and the SpMV's rate as a fraction of it.
  sustained    = the same clean loop over 4800 steps in one go, run before the K-step windows (informational; `value` stays the
                 K-step figure): after idle time the device runs the loop ~5 % slower for its first 50-150 ms under load, on
                 fixed memory (profiles/r04_placement_lab9.txt); this leg absorbs that.
  phases_us    = per-kernel breakdown of a loop body from an event after every launch (a separate pass).
  K < 100      : the K-step timing is repeated and the MEDIAN is reported (timed_repeats).
  ok           = false when the line is a degraded one (below) or a pre-flight failure.

`python bench.py --gpus N` works as typed: the parent process starts N worker processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, free port) BEFORE anything touches the GPU, relays rank 0's
single JSON line and any non-zero exit code, and never initialises HIP itself.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already exist and each
rank process supervises ONE worker the same way.

  degraded     = (N > 1) the communicator's data plane is validated and timed FIRST; behind that checkpoint a failure of
                 the peer-mapped plane -- wrong values in its pre-flight, a crash, a time-out -- does not lose the run: the
                 line is then quoted on the communicator's plane (config.data_plane says which), carries "ok": false and a
                 `degraded` block (what failed, exit codes, stderr tails, the files holding every worker's whole stderr) and
                 the exit code is 0.  A failure of the communicator plane's own pre-flight, or of anything before the
                 checkpoint, prints no rate and exits non-zero.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sparsebench_amd.bench.context import host_cores, quiet_stdout  # noqa: E402  (no GPU library is loaded by these imports)
from sparsebench_amd.bench.line import pmc_traffic  # noqa: E402,F401  (re-exported: tests, tools)
from sparsebench_amd.bench.preflight import check_history, golden_key  # noqa: E402,F401
from sparsebench_amd.bench.supervisor import supervise  # noqa: E402


# ------------------------------------------------------------------------------------------------
# CPU baseline (child processes, never load the GPU libraries).  These legs are the only code outside tests/ and
# __graft_entry__.smoke() that touches oracle/: they stay in this file, outside the product package.
# ------------------------------------------------------------------------------------------------
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
    why = "oracle/_ref/libsbref_crs_omp.so is not in this tree (the reference build did not travel)"
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
            # (small problems: both runs are dominated by set-up noise and the difference can come out <= 0)
            if k2 > 0 and t2 - t1 > 0:
                return {"value": k2 / (t2 - t1), "unit": "iterations/s", "cores": threads, "kind": "reference",
                        "sample": sample.replace("difference of two solveCG runs: set-up, prologue and residual check cancel",
                                                 "ONE solveCG run incl. its prologue and residual check: the two-run difference "
                                                 "was not positive at this size (%.3f s vs %.3f s)" % (t1 - t0, t2 - t1))}
            why = "the reference's solveCG returned no iterations (k = %d, %d)" % (k1, k2)
    except Exception as e:  # fall through to the port
        why = "the reference leg failed: %s" % e
    sys.stderr.write("cpu_baseline: kind 'port' because %s\n" % why)
    with quiet_stdout():
        g = po.GMatrix.generate(n, n, n)
        thr = ctypes.c_int(0)
        dt = po.lib().orc_time_cg_iters(g.ptr, iters, ctypes.byref(thr))
    return {"value": iters / dt, "unit": "iterations/s", "cores": thr.value, "kind": "port", "why_port": why,
            "sample": "HPCG %d^3 CRS, %d CG loop bodies of the oracle's OpenMP restatement, 1 rank x %d threads" % (n, iters, thr.value)}


def cpu_mpi_leg(n, iters, cores, rate_hint=0.0):
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
            # (the reference prints its loop time with two decimals: run long enough -- >= ~2 s at the OpenMP-only rate --
            #  for that resolution not to matter; 64^3 takes 0.2 ms per iteration)
            a = run(max(5 * iters, int(2.0 * rate_hint)) + 1)
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
            mpi = cpu_mpi_leg(n, iters, cores, rate_hint=res["value"])
        except Exception as e:
            mpi = None
            sys.stderr.write("cpu_baseline: MPI leg failed (%s)\n" % e)
        res["mpi_openmp"] = mpi
        if mpi and mpi["value"] > res["value"]:
            res.update(value=mpi["value"], cores=mpi["cores"], sample=mpi["sample"])
    return res


def run_rank(args):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    irregular = args.workload == "irregular"
    n = args.n if args.n > 0 else (80 if irregular else 128)
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(args.workload, n, args.cpu_iters)  # before the GPU is initialised
    from sparsebench_amd.bench import context, hpcg, irregular as irregular_wl
    ctx = context.RankContext(args)
    out = (irregular_wl if irregular else hpcg).run(ctx, cpu)
    if ctx.rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    ctx.finalize()


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
                    help="sb_cg_set_fused level: 1 five launches per loop body (default); 0 the reference's op list; "
                         "2 / 3 the measured-slower alternatives (lab builds only)")
    ap.add_argument("--fuse-p", type=int, default=-1, help="the p update inside the SpMV launch where the matrix allows it: 1 / 0, "
                                                           "-1 (default): the library's choice")
    ap.add_argument("--fuse-alpha", type=int, default=-1, help="the alpha step inside the r update's launch (one rank): 1 / 0, -1 (default): "
                                                               "the library's choice")
    ap.add_argument("--fuse-beta", type=int, default=-1, help="the beta step at the head of the p update where that is a launch of its own: "
                                                              "1 / 0, -1 (default): the library's choice")
    ap.add_argument("--pack-mode", type=int, default=-1,
                    help="SpMV stream: 0 reference layout, 5 masked row programs + LDS x-window where the matrix qualifies "
                         "(default -1: the library's choice); 1-3 intermediate forms (lab builds only)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="N > 1 data plane: rccl (production) or host (gloo-staged; lets N ranks share one GPU "
                         "to rehearse the multi-rank flow -- its numbers are not a benchmark)")
    ap.add_argument("--loops", default="both", choices=["both", "reference", "structure"],
                    help="which CG loops to time: both (default: `value` on the reference-layout SpMV, the compressed-mirror loop in "
                         "`structure_exploiting`), reference (only the former), structure (lab use: only the latter; the line says so)")
    ap.add_argument("--no-rccl-leg", action="store_true", help="N > 1: do not time the second data plane (rccl_only)")
    ap.add_argument("--no-push-inside-leg", action="store_true", help="N > 1: do not time the push-inside-the-SpMV variant")
    ap.add_argument("--no-preflight", action="store_true", help="skip the known-answer checks (lab use; the line says so)")
    ap.add_argument("--passes", type=lambda v: set(v.split(",")), default={"clean", "events", "phases"},
                    help="which timed passes to run besides the clean one: events (HIP events around every SpMV launch: the roofline "
                         "leg), phases (an event after every launch: the per-kernel breakdown).  `--passes clean` under rocprofv3 "
                         "profiles exactly the loop `value` is quoted on")
    ap.add_argument("--sustained-steps", type=int, default=4800, help="also time the clean loop over this many steps in one go "
                                                                       "(reported as `sustained`; 0: off)")
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
        return supervise(list(range(args.gpus)), args.gpus, sys.argv[1:], True)  # before any HIP call; this process stays off the GPU
    if args.gpus == 1 and world > 1:
        sys.stderr.write("bench: --gpus 1 but WORLD_SIZE=%d\n" % world)
        return 2
    if world > 1 and not os.environ.get("SB_BENCH_RANK_PROCESS"):
        # a rank process started by torch.distributed.run: supervise ONE worker (this process stays off the GPU)
        return supervise([int(os.environ.get("RANK", "0"))], world, sys.argv[1:], False)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
