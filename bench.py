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
  roofline_reference_layout also carries a CLEAN (event-free) cg_iterations_per_s of the loop run on that
                 kernel: `value` exploits the structure of the matrix (compressed mirror), that rate does not.
  cpu_baseline = the reference's own solveCG (oracle/_ref, upstream flags + OpenMP) timed on this box's
                 host cores on a bounded sample (rank 0, N=1)
  preflight    = before anything is timed, every rank solves 20 CG iterations on a 32^3-per-rank problem and on
                 the bench's own bricks and checks the history against closed-form known answers (r.r of the
                 prologue, p.Ap of the first body: exact integers at any size and rank count,
                 sparsebench_amd/knownanswers.py), against the committed oracle histories in the GPU's dot order
                 (tests/golden/cg_hist_tree.json: bit for bit), and against every other rank's history (identical
                 bits).  Any mismatch: no rate is printed from that data plane (one plane only: exit code 4).  At N > 1
                 this runs on BOTH data planes, each before it is timed.
                 (irregular workload: the stand-in at 24^3 nodes against the history the reference itself produced on it,
                 every format within 1e-12, CRS and Sell-64-1 identical bits.)
  rccl_only    = (N > 1) the same K steps timed again with the peer-mapped paths switched off
                 (sb_comm_data_plane(0): RCCL all-reduce + send/recv), so one invocation yields both curves.
  sustained    = the same clean loop over 4800 steps in one go (informational: a K = 20 window is 1 ms of GPU work between
                 host-side pauses, and the rate of a long run is a few per cent higher; `value` stays the K-step figure).
  phases_us    = per-kernel breakdown of a loop body from an event after every launch (a separate pass).
  K < 100      : the K-step timing is repeated and the MEDIAN is reported (timed_repeats).

`python bench.py --gpus N` works as typed: the parent process starts N worker processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, free port) BEFORE anything touches the GPU, relays rank 0's
single JSON line and any non-zero exit code, and never initialises HIP itself.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already exist and each
rank process supervises ONE worker the same way.

  degraded     = (N > 1) the communicator's data plane is validated and timed FIRST; behind that checkpoint a failure of
                 the peer-mapped plane -- wrong values in its pre-flight, a crash, a time-out -- does not lose the run: the
                 line is then quoted on the communicator's plane (config.data_plane says which), carries a `degraded`
                 block (what failed, exit codes, stderr tails) and the exit code is 0.  A failure of the communicator
                 plane's own pre-flight, or of anything before the checkpoint, prints no rate and exits non-zero.
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
    THIS kernel source (content hash of sparsebench_amd/csrc/*, sparsebench_amd/srchash.py -- a kernel
    edit invalidates the entry whether or not anybody bumped a version string).  Returns (bytes, source,
    note): bytes is None -- never a stale constant -- when no pass matches, and the note says what is missing."""
    from sparsebench_amd import srchash
    version = srchash.csrc_hash()
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
        if e.get("source_hash", doc.get("source_hash")) == version:
            return e["bytes_per_launch"], "profiles/" + name, None
        stale = stale or "profiles/%s holds %s/%s for kernel source %r, the library was built from %r" % (
            name, workload, kernel, e.get("source_hash", doc.get("source_hash")), version)
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


# ------------------------------------------------------------------------------------------------
# N > 1: the parent starts the ranks itself
# ------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


MARK = "@@sbbench "  # prefix of the worker -> supervisor lines on a worker's stdout (never relayed)


def supervise(ranks, n_gpus, argv, own_env):
    """N > 1: every rank's work runs in a WORKER process under a supervisor that never touches the GPU (no HIP call, no
    exec of a process that has).  `python bench.py --gpus N`: one supervisor (this process) starts all N workers
    (own_env: it sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* on a free port).  Under torch.distributed.run every rank
    process is the supervisor of ONE worker that inherits its environment unchanged.

    Worker stdout is captured: `MARK` lines are protocol, a line starting with "{" is rank 0's JSON line, anything else
    goes to stderr.  ALL workers are polled: as soon as one exits non-zero the others -- which would otherwise sit in a
    collective that has no time-out -- are terminated.  Nothing is ever restarted or exec'd.

    Degraded completion (first contact with real xGMI links must yield a line if ANY validated data plane works): the
    workers validate and time the communicator's data plane (RCCL all-reduce + send/recv) FIRST and then announce a
    checkpoint; rank 0 hands over a provisional line quoted on that plane.  If a worker then dies, times out or
    hangs in the peer-mapped legs, the supervisor prints the provisional line with a "degraded" block (who failed, exit
    code, stderr tail) and exits 0 -- a rate from a plane that passed its pre-flight, labelled as such.  A failure
    BEFORE the checkpoint (or of the pre-flight itself) is an error: no rate, the worker's exit code."""
    import threading
    port = free_port() if own_env else None
    procs, state = {}, {}
    for r in ranks:
        env = dict(os.environ, SB_BENCH_RANK_PROCESS="1")
        if own_env:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        script = os.environ.get("SB_BENCH_WORKER_SCRIPT", os.path.abspath(__file__))  # (test hook: tests/test_bench_supervisor.py)
        procs[r] = subprocess.Popen([sys.executable, script] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        state[r] = {"checkpoint": None, "provisional": None, "final": None, "err": []}

    def read_out(r):
        for raw in procs[r].stdout:
            ln = raw.decode(errors="replace").rstrip("\n")
            if ln.startswith(MARK + "checkpoint"):
                state[r]["checkpoint"] = time.time()
            elif ln.startswith(MARK + "provisional "):
                state[r]["provisional"] = ln[len(MARK + "provisional "):]
            elif ln.startswith("{"):
                state[r]["final"] = ln
            elif ln.strip():
                sys.stderr.write(ln + "\n")

    def read_err(r):
        for raw in procs[r].stderr:
            ln = raw.decode(errors="replace")
            sys.stderr.write(ln)
            state[r]["err"] = (state[r]["err"] + [ln.rstrip("\n")])[-12:]

    threads = [threading.Thread(target=f, args=(r,), daemon=True) for r in ranks for f in (read_out, read_err)]
    for t in threads:
        t.start()
    # bounds: the whole run, and the legs behind the checkpoint (seconds of work when healthy)
    t_total = float(os.environ.get("SB_BENCH_TIMEOUT_S", "1500"))
    t_after = float(os.environ.get("SB_BENCH_AFTER_CHECKPOINT_S", "240"))
    t0 = time.time()
    rc, failed, why = 0, None, None
    # (a supervisor that is told to stop -- torch.distributed.run ends the group when one rank fails -- ends its workers)
    import signal
    stop = {"sig": None}
    old = {sg: signal.signal(sg, lambda n, f: stop.__setitem__("sig", n)) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        while True:
            codes = {r: p.poll() for r, p in procs.items()}
            bad = [(r, c) for r, c in codes.items() if c not in (None, 0)]
            if bad:
                failed, rc = bad[0]
                why = "rank %d exited with code %d" % (failed, rc)
                break
            if all(c == 0 for c in codes.values()):
                break
            now = time.time()
            cps = [state[r]["checkpoint"] for r in ranks]
            if stop["sig"] is not None:
                failed, rc, why = -1, 128 + stop["sig"], "the supervisor received signal %d" % stop["sig"]
                break
            if now - t0 > t_total:
                failed, rc, why = -1, 124, "the run exceeded SB_BENCH_TIMEOUT_S = %.0f s" % t_total
                break
            if all(cps) and now - max(cps) > t_after:
                failed, rc, why = -1, 124, "the legs behind the checkpoint exceeded SB_BENCH_AFTER_CHECKPOINT_S = %.0f s" % t_after
                break
            time.sleep(0.05)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
        if failed is not None:
            sys.stderr.write("bench: %s; ending the other ranks\n" % why)
            time.sleep(1.0)  # (let ranks that are failing for the same reason print their own message)
        for p in procs.values():  # end exactly the processes we started
            if p.poll() is None:
                p.terminate()
        for p in procs.values():
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    for t in threads:
        t.join(timeout=10)
    mine0 = 0 in state
    if failed is None:
        if mine0:
            if state[0]["final"]:
                print(state[0]["final"], flush=True)
            else:
                rc = 3
                sys.stderr.write("bench: rank 0 printed no JSON line\n")
        return rc
    if stop["sig"] is None and all(state[r]["checkpoint"] for r in ranks):
        # behind the checkpoint: the communicator's plane was validated and timed on every rank
        if mine0:
            if state[0]["final"]:  # (rank 0 had finished; somebody else failed on the way out)
                print(state[0]["final"], flush=True)
            elif state[0]["provisional"]:
                line = json.loads(state[0]["provisional"])
                tails = {str(r): state[r]["err"][-6:] for r in ranks if procs[r].returncode not in (0, None, -15)}
                line["degraded"] = {"why": why + " in the legs behind the communicator-plane checkpoint",
                                    "value_is_quoted_on": line["config"].get("data_plane"),
                                    "exit_codes": {str(r): procs[r].returncode for r in ranks}, "stderr_tail": tails}
                print(json.dumps(line), flush=True)
            else:
                sys.stderr.write("bench: checkpoint without a provisional line\n")
                return rc
        sys.stderr.write("bench: DEGRADED completion (%s): the line is quoted on the communicator's data plane\n" % why)
        return 0
    if mine0 and state[0]["final"]:  # a failure line (pre-flight): relay it, keep the exit code
        print(state[0]["final"], flush=True)
    return rc


# ------------------------------------------------------------------------------------------------
# pre-flight: known answers before anything is timed
# ------------------------------------------------------------------------------------------------
PREFLIGHT_ITERS = 20   # itermax of the pre-flight solves (and of the committed P-rank goldens)
PREFLIGHT_SMALL = 32   # brick edge of the small pre-flight problem


def golden_key(n, P, fmt, Cc, sigma):
    return "hpcg%d_x%d_%s" % (n, P, "crs" if fmt == "crs" else "scs_C%d_sigma%d" % (Cc, sigma))


def load_goldens():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "cg_hist_tree.json")))
    except (OSError, ValueError):
        return {}


def check_history(label, rr, pap, n, world, key, goldens):
    """One pre-flight solve against what is known about it.  Returns (record, problems)."""
    import numpy as np
    from sparsebench_amd import knownanswers as ka
    rec = {"case": label, "iterations": int(len(pap))}
    bad = []
    want_rr0, want_pap1 = float(ka.hpcg_rr0(n, n, n * world)), float(ka.hpcg_pAp1(n, n, n * world))
    if len(rr) < 2 or len(pap) < 1:
        return rec, ["%s: the solve produced no history (%d r.r, %d p.Ap values)" % (label, len(rr), len(pap))]
    rec["rr0"], rec["rr0_closed_form"] = float(rr[0]), want_rr0
    rec["pAp1"], rec["pAp1_closed_form"] = float(pap[0]), want_pap1
    if rr[0] != want_rr0:
        bad.append("%s: r.r of the prologue is %.17g, closed form %.17g" % (label, rr[0], want_rr0))
    if pap[0] != want_pap1:
        bad.append("%s: p.Ap of the first body is %.17g, closed form %.17g (first product that needs the halo)" % (label, pap[0], want_pap1))
    if not (np.all(np.isfinite(rr)) and np.all(np.isfinite(pap)) and np.all(pap > 0) and np.all(rr > 0)):
        bad.append("%s: the history holds non-finite or non-positive values" % label)
    g = goldens.get(key)
    rec["golden"] = key if g else None
    if g:
        grr = np.array([float(v) for v in g["rr"]])
        gpa = np.array([float(v) for v in g["pAp"]])
        m, q = min(len(grr), len(rr)), min(len(gpa), len(pap))
        if m < PREFLIGHT_ITERS - 2 or q < PREFLIGHT_ITERS - 2:
            bad.append("%s: only %d / %d values to compare with the golden history" % (label, m, q))
        elif not (np.array_equal(rr[:m], grr[:m]) and np.array_equal(pap[:q], gpa[:q])):
            d = np.nonzero(rr[:m] != grr[:m])[0]
            e = np.nonzero(pap[:q] != gpa[:q])[0]
            bad.append("%s: history differs from tests/golden/cg_hist_tree.json[%s]: first r.r mismatch at %s, first p.Ap "
                       "mismatch at %s" % (label, key, d[0] if len(d) else None, e[0] if len(e) else None))
        rec["golden_values_compared"] = int(m + q)
    return rec, bad


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def kernel_name(fmt, mode, crs_split=True):
    native = ("spmv_crs_split" if crs_split else "spmv_crs_stream") if fmt == "crs" else "spmv_scs64"
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
    import numpy as np
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

    if world > 1:  # set-up (generator, partitioner, layout) is OpenMP-parallel on the host: share the cores between the ranks
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, host_cores()[1] // world)))
    if os.environ.get("SB_BENCH_TEST_DIE_RANK") == str(rank):  # test hook: a rank that dies before the first collective
        sys.stderr.write("bench: rank %d: SB_BENCH_TEST_DIE_RANK is set, exiting with code 7 (test hook)\n" % rank)
        os._exit(7)

    from sparsebench_amd import capi, hostapi
    capi.load()
    ndev = capi.load().sb_device_count()
    device = local % ndev if args.transport == "host" and ndev > 0 else local
    L = capi.init(device)
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

    def gather(obj):
        """every rank's `obj`, in rank order"""
        if dist is None:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    def rank_max(v):
        if dist is None:
            return float(v)
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    K, W = args.steps, args.warmup
    repeats = 1 if K >= 100 else 9  # a 1 ms window moves by a few per cent from run to run: median of 9
    vphase, launches, collectives, fuse_p = 0, 5, 0, 0

    # ---- pre-flight ------------------------------------------------------------------------------------
    goldens = load_goldens()

    def preflight(plane_name, prob_full):
        """20 CG iterations on a 32^3-per-rank problem and on the bench's own bricks, on the data plane that is
        selected right now.  Collective; returns (records, problems) identical on every rank."""
        import hashlib
        records, problems = [], []
        cases = [("%s: 32^3 per rank" % plane_name, PREFLIGHT_SMALL, None)]
        if n != PREFLIGHT_SMALL:
            cases.append(("%s: bench bricks (%d^3 per rank)" % (plane_name, n), n, prob_full))
        else:
            cases = [("%s: bench bricks (32^3 per rank)" % plane_name, n, prob_full)]
        for label, nn, pr in cases:
            own = pr is None
            if own:
                with quiet_stdout():
                    pr = hostapi.Problem("generate", nn, nn, nn, fmt=args.fmt, Cc=args.C, sigma=args.sigma, rank=rank, size=world)
            cg = hostapi.CG(pr, fused=args.fused, graph=False, fuse_p=args.fuse_p, fuse_alpha=args.fuse_alpha, fuse_beta=args.fuse_beta)
            cg.solve(PREFLIGHT_ITERS, 0.0)
            rr, pap = cg.history()
            cg.free()
            if own:
                pr.free()
            rec, bad = check_history(label, rr, pap, nn, world, golden_key(nn, world, args.fmt, args.C, args.sigma), goldens)
            digest = hashlib.sha256(rr.tobytes() + pap.tobytes()).hexdigest()[:16]
            everyone = gather((digest, bad))
            digests = [d for d, _ in everyone]
            rec["history_sha256_by_rank"] = digests
            if len(set(digests)) != 1:
                bad = bad + ["%s: the ranks hold DIFFERENT histories (all-reduced scalars must be identical bits): %s" % (label, digests)]
            for r, (_, b) in enumerate(everyone):  # a problem seen by any rank is everybody's problem
                for msg in b:
                    if msg not in bad:
                        bad.append("rank %d: %s" % (r, msg))
            rec["ok"] = not bad
            records.append(rec)
            problems += bad
        return records, problems

    def fail_preflight(records, problems, workload):
        if rank == 0:
            for msg in problems:
                sys.stderr.write("bench: PRE-FLIGHT FAILED: %s\n" % msg)
            print(json.dumps({"metric": "cg_iterations_per_s", "value": None, "unit": "iterations/s", "n_gpus": world,
                              "steps": K, "warmup": W, "error": "pre-flight check failed: nothing was timed",
                              "config": {"workload": workload},
                              "preflight": {"ok": False, "problems": problems, "checks": records}}), flush=True)
        if world > 1:
            L.sb_sync()
            dist.barrier()
        # (no sb_comm_finalize / destroy_process_group: the run is invalid, leave at once with the failure code)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(4)

    def measure(prob, modes, clean_all=False, phases=True, sustained=False):
        """timed passes on one resident matrix.  modes: kernel modes to time with per-launch events; the
        first is the one `value` is quoted on.  Every mode gets a clean pass (no events) when clean_all."""
        cg = hostapi.CG(prob, fused=args.fused, graph=bool(args.graph), fuse_p=args.fuse_p, fuse_alpha=args.fuse_alpha, fuse_beta=args.fuse_beta)
        nonlocal vphase, launches, collectives, fuse_p
        vphase, launches, collectives, fuse_p = cg.vector_phase(), cg.launches_per_body(), cg.collectives_per_body(), cg.fuse_p()

        def timed_pass(with_spmv_events, with_phases=False, steps=None):
            """exactly K loop bodies (`steps`: the sustained leg's count), in segments restarted from x0 = 0 outside the clock"""
            total, left, spmv_ms, spmv_n = 0.0, (steps or K), 0.0, 0
            phase_acc = {}
            while left > 0:
                seg = min(left, SEGMENT)
                cg.spmv_timing(False)
                cg.phase_timing(False)
                cg.start(itermax=W + 2 + seg, eps=0.0)  # prologue
                cg.run_iters(W + 1)                     # warm-up bodies, untimed
                before = cg.counters()
                cg.spmv_timing(with_spmv_events)
                cg.phase_timing(with_phases)
                barrier()
                t0 = time.perf_counter()
                cg.run_iters(seg)
                L.sb_sync()  # this rank's K steps are complete on its GPU ...
                dt = time.perf_counter() - t0
                barrier()    # ... and nobody moves on before all are (the max over ranks is taken below;
                #                 the gloo TCP barrier itself is control plane, not part of a CG step)
                after = cg.counters()
                if with_phases:
                    for name, (us, cnt) in cg.phase_us().items():
                        a = phase_acc.setdefault(name, [0.0, 0])
                        a[0] += us * cnt
                        a[1] += cnt
                cg.phase_timing(False)
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
            return total, spmv_ms, spmv_n, {k: (v[0] / v[1], v[1] // max(1, steps or K)) for k, v in phase_acc.items() if v[1]}

        res = {}
        for i, mode in enumerate(modes):
            got = prob.use_packed(mode)
            if got != mode:
                continue
            t_clean = t_mine = None
            all_reps = []
            if i == 0 or clean_all:
                mine, agreed = [], []
                for _ in range(repeats):
                    dt = timed_pass(False)[0]
                    mine.append(dt)
                    agreed.append(rank_max(dt))  # the step ends when the slowest rank is done
                order = sorted(range(repeats), key=lambda j: agreed[j])
                mid = order[repeats // 2]        # the median repeat (the same one on every rank)
                t_clean, t_mine, all_reps = agreed[mid], mine[mid], agreed
            t_ev, ms, cnt = None, 0.0, 0
            if "events" in args.passes:
                t_ev, ms, cnt, _ = timed_pass(True)
            ph = timed_pass(False, True)[3] if phases and "phases" in args.passes and (i == 0 or clean_all) else None
            # the same loop over a run long enough for the device to settle (`sustained`): a timed window of K = 20 steps is 1 ms of
            # GPU work between host-side pauses, and the rate of a run of thousands of steps is a few per cent higher
            t_sus = None
            if sustained and (i == 0 or clean_all) and args.sustained_steps > K:
                t_sus = rank_max(timed_pass(False, steps=args.sustained_steps)[0])
            res[mode] = {"fuse_p": cg.fuse_p(), "t_clean": t_clean, "t_mine": t_mine, "t_repeats": all_reps, "t_ev": t_ev, "t_sus": t_sus,
                         "spmv_us": 1e3 * ms / max(cnt, 1), "launches": cnt, "launches_per_body": cg.launches_per_body(),
                         "moved": prob.stream_bytes(), "alg": prob.spmv_bytes(), "phases": ph}
        prob.use_packed(modes[0])
        cg.free()
        return res

    def vector_bytes(nr):
        """bytes the fused loop's vector kernels move per iteration.  Separate launches: p update (+ the x update
        owed by the previous body) 40 B/row, r update + r.r partials 24 B/row.  One-launch vector phase: r, Ap, p, x
        read and r, p, x written once: 56 B/row.  Plus the partials written and read back."""
        return (56.0 if vphase else 64.0) * nr + 2 * 8.0 * (nr / 256.0)

    def phase_table(ph):
        return {k: round(v[0], 3) for k, v in ph.items()} if ph else None

    out = None
    if not irregular:
        with quiet_stdout():
            prob = hostapi.Problem("generate", n, n, n, fmt=args.fmt, Cc=args.C, sigma=args.sigma, rank=rank, size=world)
        default = prob.use_packed(args.pack_mode) if args.pack_mode >= 0 else prob.pack_info()["mode"]
        workload = "hpcg_27pt_%d^3_per_gpu_%s_C%d_sigma%d" % (n, args.fmt, args.C, args.sigma)
        p2p_dots, p2p_halo = (L.sb_comm_p2p_enabled(), L.sb_halo_p2p_enabled(prob.halo)) if world > 1 else (0, 0)
        second_plane = world > 1 and (p2p_dots or p2p_halo) and not args.no_rccl_leg
        coll = "rccl" if args.transport == "rccl" else "host_staged_gloo"

        supervised = bool(os.environ.get("SB_BENCH_RANK_PROCESS"))

        def plane_name(plane):
            return "one GPU" if world == 1 else ("peer-mapped data plane" if plane and (p2p_dots or p2p_halo) else "%s data plane" % coll)

        def compact_line(c, plane, recs, steps_ms):
            """a complete line of the contract quoted on ONE validated data plane (the provisional line handed to the
            supervisor behind the communicator-plane checkpoint; the degraded line when the peer-mapped plane fails its pre-flight)"""
            it = K / c["t_clean"]
            return {"metric": "cg_iterations_per_s", "value": world * it,
                    "unit": "iterations/s (one iteration = one %d^3-brick CG step; summed over GPUs)" % n,
                    "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * c["t_clean"] / K,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                    "config": {"workload": workload, "rows_per_gpu": prob.nr, "nnz_per_gpu": prob.nnzTrue, "index_type": "u32",
                               "parallelism": "1d_block_row_x%d" % world, "data_plane": plane_name(plane),
                               "halo_exchange": "rccl_send_recv" if args.transport == "rccl" else "host_staged_gloo",
                               "dot_allreduce": coll, "launches_per_iteration": c["launches_per_body"],
                               "p_update_inside_spmv": bool(c["fuse_p"]), "library": version},
                    "timed_repeats": repeats, "ms_per_step_repeats": [1e3 * t / K for t in c["t_repeats"]],
                    "global_iterations_per_s": it, "phases_us": phase_table(c["phases"]),
                    "per_rank": {"ms_per_step": steps_ms}, "preflight": {"ok": True, "checks": recs},
                    "device": L.sb_device_name().decode(), "cpu_baseline": None}

        # pre-flight on every data plane that will be timed, before that plane is timed.  N > 1 with both planes up: the
        # communicator's plane FIRST (validated, timed, handed to the supervisor as a provisional line: the checkpoint), so
        # that a failure of the peer-mapped plane on its first contact with real links still leaves a validated rate.
        checks, problems = [], []
        res_coll, degraded = None, None
        coll_problems = []
        if second_plane:
            L.sb_comm_data_plane(0)
            if not args.no_preflight:
                checks, coll_problems = preflight(plane_name(0), prob)
            if coll_problems:
                # the communicator's plane gives wrong results here: nothing is timed on it, no checkpoint; the run goes on to the
                # peer-mapped plane, whose own pre-flight decides whether there is a rate at all (both wrong: exit code 4)
                if rank == 0:
                    for msg in coll_problems:
                        sys.stderr.write("bench: PRE-FLIGHT FAILED on the communicator's data plane: %s\n" % msg)
                L.sb_comm_data_plane(1)
            else:
                res_coll = measure(prob, [default])
                launches_coll, collectives_coll = launches, collectives
                L.sb_comm_data_plane(1)
                coll_ms = gather(1e3 * res_coll[default]["t_mine"] / K)
                provisional = compact_line(res_coll[default], 0, list(checks), coll_ms) if rank == 0 else None
                if supervised:
                    if rank == 0:
                        print(MARK + "provisional " + json.dumps(provisional), flush=True)
                    barrier()
                    print(MARK + "checkpoint", flush=True)
                if os.environ.get("SB_BENCH_TEST_DIE_AFTER_CHECKPOINT") == str(rank):  # test hook: a crash in the peer-mapped legs
                    sys.stderr.write("bench: rank %d: SB_BENCH_TEST_DIE_AFTER_CHECKPOINT is set, exiting with code 9 (test hook)\n" % rank)
                    os._exit(9)
        if not args.no_preflight:
            recs, bad = preflight(plane_name(1), prob)
            checks += recs
            if bad and coll_problems:
                fail_preflight(checks, coll_problems + bad, workload)
            if bad and second_plane:
                # the peer-mapped plane gives WRONG results here, the communicator's plane passed: no rate from the former, the
                # line is quoted on the latter and says so (exit code 0: a validated rate; the failure is in the line and on stderr)
                degraded = bad
            elif bad:
                fail_preflight(checks, bad, workload)
        if degraded:
            if rank == 0:
                for msg in degraded:
                    sys.stderr.write("bench: PRE-FLIGHT FAILED on the peer-mapped data plane: %s\n" % msg)
                sys.stderr.write("bench: DEGRADED: the line is quoted on the communicator's data plane, which passed\n")
                provisional["degraded"] = {"why": "the peer-mapped data plane failed its pre-flight; nothing was timed on it",
                                           "value_is_quoted_on": provisional["config"]["data_plane"], "problems": degraded}
                provisional["preflight"] = {"ok": False, "checks": checks, "problems": degraded,
                                            "ok_on_the_plane_value_is_quoted_on": True}
                print(json.dumps(provisional), flush=True)
            L.sb_comm_data_plane(0)
            barrier()
            prob.free()
            L.sb_comm_finalize()
            dist.destroy_process_group()
            return

        modes = [default] + ([0] if default != 0 else [])
        res = measure(prob, modes, clean_all=(world == 1) or args.all_clean, sustained=True)
        # third leg, peer-mapped halo only: the halo push inside the SpMV launch (one launch fewer per body).  Which
        # variant is faster can only be decided with one rank per GPU, i.e. by this very run on a real node; validated by
        # its own pre-flight, and a failure here does not invalidate `value` (the variant is simply reported as failed).
        res_inside, inside_checks, inside_problems = None, [], []
        if world > 1 and p2p_halo and default >= 3 and not args.no_push_inside_leg:
            L.sb_comm_halo_push_inside(1)
            if not args.no_preflight:
                inside_checks, inside_problems = preflight("peer-mapped data plane, push inside the SpMV launch", prob)
            if not inside_problems:
                res_inside = measure(prob, [default])
                launches_inside = launches
            L.sb_comm_halo_push_inside(0)
            cg_tmp = hostapi.CG(prob, fused=args.fused, fuse_p=args.fuse_p, fuse_alpha=args.fuse_alpha, fuse_beta=args.fuse_beta)
            launches, vphase, collectives, fuse_p = cg_tmp.launches_per_body(), cg_tmp.vector_phase(), cg_tmp.collectives_per_body(), cg_tmp.fuse_p()
            cg_tmp.free()
        rccl = (ctypes.c_int * 3)()
        has_rccl = L.sb_comm_rccl_info(rccl) if world > 1 else 0
        per_rank = gather({"rank": rank, "device": device, "spmv_mode": default, "ms_per_step": 1e3 * res[default]["t_mine"] / K,
                           "rccl": list(rccl) if has_rccl else None, "phases_us": phase_table(res[default]["phases"]),
                           "ms_per_step_rccl_only": (1e3 * res_coll[default]["t_mine"] / K) if res_coll else None,
                           "phases_us_rccl_only": phase_table(res_coll[default]["phases"]) if res_coll else None,
                           "ms_per_step_push_inside": (1e3 * res_inside[default]["t_mine"] / K) if res_inside else None,
                           "phases_us_push_inside": phase_table(res_inside[default]["phases"]) if res_inside else None})
        if rank == 0:
            d = res[default]
            it_s = K / d["t_clean"]
            kern = kernel_name(args.fmt, default, bool(L.sb_matrix_crs_kernel(prob.matrix)))
            if d["fuse_p"]:
                # the SpMV launch also takes the p update: + r and p_old read, p_new written, x read and written = 40 B/row,
                # the same 64 B/row of vector traffic per iteration as with the separate kernel
                kern = "spmv_prog_fusep"
                d["moved"] += 40.0 * prob.nr
                d["alg"] += 40.0 * prob.nr  # (the launch's operations: the SpMV of SURVEY 8d + the fused p / x update)
            tr = pmc_traffic(workload, kern, version) if world == 1 else (None, None, "N > 1")
            cg_moved = d["moved"] + vector_bytes(prob.nr) - (40.0 * prob.nr if d["fuse_p"] else 0.0)
            cg_alg = 96.0 * prob.nr + d["alg"] - (40.0 * prob.nr if d["fuse_p"] else 0.0)  # reference's unfused op list on its own layout
            steps_ms = [r["ms_per_step"] for r in per_rank]
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
                           "halo_exchange": ("none" if world == 1 else "peer_mapped_push_pull" if p2p_halo
                                             else "rccl_send_recv" if args.transport == "rccl" else "host_staged_gloo"),
                           "dot_allreduce": ("none" if world == 1 else "in_kernel_peer_mapped" if p2p_dots
                                             else "rccl" if args.transport == "rccl" else "host_staged_gloo"),
                           "dot_allreduce_reason": (L.sb_comm_p2p_reason().decode() if world > 1 else None),
                           "halo_exchange_reason": (L.sb_halo_p2p_reason(prob.halo).decode() if world > 1 else None),
                           "rccl_ranks": (rccl[0] if has_rccl else None),
                           "spmv_x_staging": ("lds_window" if default >= 2 else "l1_l2_gather (LDS staging measured neutral at 12 B/element)"),
                           "fused_dots": True, "vector_phase_one_launch": bool(vphase), "launches_per_iteration": launches,
                           "collective_calls_per_iteration": collectives,
                           "p_update_inside_spmv": bool(d["fuse_p"]),
                           "spmv_kernel_mode_by_rank": [r["spmv_mode"] for r in per_rank],
                           "device_by_rank": [r["device"] for r in per_rank],
                           "hip_graph": bool(args.graph), "library": version},
                "timed_repeats": repeats,
                "ms_per_step_repeats": [1e3 * t / K for t in d["t_repeats"]],
                "global_iterations_per_s": it_s,
                "roofline": roofline_block(kern, d["moved"], d["alg"], d["spmv_us"], d["launches"], *tr),
                "algorithmic_speedup": d["alg"] / d["moved"],
                "cg_moved_bytes_per_iteration": cg_moved,
                "cg_moved_GBs_per_gpu": cg_moved * it_s / 1e9,
                "cg_frac_of_hbm_peak": cg_moved * it_s / 1e9 / HBM_PEAK_GBS,
                "cg_reference_oplist_bytes_per_iteration": cg_alg,
                "ms_per_step_with_events": (1e3 * d["t_ev"] / K) if d["t_ev"] else None,
                "phases_us": phase_table(d["phases"]),
                "preflight": ({"ok": not coll_problems, "checks": checks, **({"problems": coll_problems, "ok_on_the_plane_value_is_quoted_on": True}
                                                                              if coll_problems else {})}
                              if not args.no_preflight else {"ok": None, "skipped": "--no-preflight"}),
                "compression": prob.pack_info(),
                "device": L.sb_device_name().decode(),
                "parity": {
                    "checked_in_this_run": "pre-flight histories: closed forms exact, committed oracle histories bit for bit, all ranks identical",
                    "bit_identical_to": "the pinned CPU oracle under the GPU's fixed dot order (tests/golden/cg_hist_tree.json; -m gpu tests at 64^3 / 128^3)",
                    "vs_cg_with_exactly_rounded_dots": "<= 1e-12 relative per iteration at 128^3 (observed 2.2e-14; tests/golden/cg_hist_exact.json)",
                    "vs_reference_cpu_history": "<= 1e-12 on 8^3..32^3 and on the irregular stand-in; at 64^3 / 128^3 bounded at 5e-11 / 6.5e-10: the "
                                                "reference's own sequential ddot is 2.5e-11 / 3.2e-10 away from the exactly rounded history (its "
                                                "summation error grows with n; no parallel order can follow it) -- north_star's 1e-12 is met against "
                                                "the exact history at the benchmark size, not against the reference's rounding"},
            }
            if d.get("t_sus"):
                ks = args.sustained_steps
                out["sustained"] = {"steps": ks, "value": world * ks / d["t_sus"], "ms_per_step": 1e3 * d["t_sus"] / ks,
                                    "note": "the same loop, clean, over %d steps in one go (informational; `value` is the K-step figure the "
                                            "contract asks for): a window of K = %d steps is a short burst between host-side pauses" % (ks, K)}
                if default != 0 and 0 in res and res[0].get("t_sus"):
                    out["sustained"]["reference_layout_value"] = world * ks / res[0]["t_sus"]
            if world > 1:
                out["per_rank"] = {"ms_per_step": steps_ms, "ms_per_step_min": min(steps_ms), "ms_per_step_max": max(steps_ms),
                                   "device": [r["device"] for r in per_rank], "rccl": [r["rccl"] for r in per_rank],
                                   "phases_us": [r["phases_us"] for r in per_rank]}
                ph_all = [r["phases_us"] for r in per_rank if r["phases_us"]]
                if ph_all:
                    out["phases_us_max_over_ranks"] = {k: max(p.get(k, 0.0) for p in ph_all) for k in ph_all[0]}
            out["roofline"]["note"] = (
                "bytes = what this kernel streams (lossless compressed mirror" + (" + the p update's r, p, x: the launch takes p = r + beta p and "
                "x += alpha p too" if d["fuse_p"] else "") + ", %.1f MB instead of the reference layout's "
                "%.1f MB): a fraction of the HBM peak on MOVED bytes -- this kernel exploits the structure of the matrix (repeating row "
                "shapes), and at this size its whole working set (mirror + five vectors) stays in the 256 MiB Infinity Cache across "
                "iterations (profiles/r03_mall_lab.txt), so it is bound by per-tile latency chains, not by HBM; the HBM-roofline figure "
                "of SURVEY 8d belongs to the kernel that streams the reference's arrays, and to the loop run on it: "
                "roofline_reference_layout" % (
                    d["moved"] / 1e6, d["alg"] / 1e6) if default > 0 else "kernel streams the reference layout: bytes = SURVEY 8d")
            if default != 0 and 0 in res:
                r0 = res[0]
                k0 = kernel_name(args.fmt, 0, bool(L.sb_matrix_crs_kernel(prob.matrix)))
                tr0 = pmc_traffic(workload, k0, version) if world == 1 else (None, None, "N > 1")
                blk = roofline_block(k0, r0["moved"], r0["alg"], r0["spmv_us"], r0["launches"], *tr0)
                blk["cg_iterations_per_s_with_events"] = (world * K / r0["t_ev"]) if r0["t_ev"] else None
                if r0["t_clean"]:
                    blk["cg_iterations_per_s"] = world * K / r0["t_clean"]
                    blk["ms_per_step"] = 1e3 * r0["t_clean"] / K
                    cgm = r0["moved"] + vector_bytes(prob.nr)
                    blk["cg_frac_of_hbm_peak"] = cgm * (K / r0["t_clean"]) / 1e9 / HBM_PEAK_GBS
                    blk["phases_us"] = phase_table(r0["phases"])
                blk["note"] = ("the loop with the SpMV streaming the reference's own Sell-C-sigma / CRS arrays (12 B per stored element): "
                               "bytes = SURVEY 8d's algorithmic figure, no use of the matrix's structure; `value` is the same loop on the "
                               "compressed mirror")
                out["roofline_reference_layout"] = blk
                # both rates with equal standing: `value` is the first (the library's default kernel choice)
                out["cg_iterations_per_s_by_spmv_kernel"] = {
                    kern + " (lossless compressed mirror: exploits the matrix's repeating row shapes)": world * it_s,
                    k0 + " (streams the reference's arrays: SURVEY 8d bytes, no use of structure)": blk.get("cg_iterations_per_s")}
            if res_coll:
                c = res_coll[default]
                cm = [r["ms_per_step_rccl_only"] for r in per_rank]
                out["rccl_only"] = {
                    "value": world * K / c["t_clean"], "ms_per_step": 1e3 * c["t_clean"] / K,
                    "ms_per_step_repeats": [1e3 * t / K for t in c["t_repeats"]],
                    "halo_exchange": "rccl_send_recv" if args.transport == "rccl" else "host_staged_gloo",
                    "dot_allreduce": coll, "launches_per_iteration": launches_coll, "collective_calls_per_iteration": collectives_coll,
                    "per_rank_ms_per_step": cm, "phases_us": phase_table(c["phases"]),
                    "phases_us_by_rank": [r["phases_us_rccl_only"] for r in per_rank],
                    "note": "same bricks, same K steps, peer-mapped paths switched off (sb_comm_data_plane(0)): the communicator's "
                            "all-reduce and send/recv carry the dots and the halo"}
            if res_inside:
                c = res_inside[default]
                out["push_inside"] = {
                    "value": world * K / c["t_clean"], "ms_per_step": 1e3 * c["t_clean"] / K,
                    "ms_per_step_repeats": [1e3 * t / K for t in c["t_repeats"]], "launches_per_iteration": launches_inside,
                    "per_rank_ms_per_step": [r["ms_per_step_push_inside"] for r in per_rank], "phases_us": phase_table(c["phases"]),
                    "phases_us_by_rank": [r["phases_us_push_inside"] for r in per_rank],
                    "preflight": {"ok": True, "checks": inside_checks},
                    "note": "the same K steps with the rank's halo push carried by the first workgroups of the SpMV launch "
                            "(sb_comm_halo_push_inside(1)) instead of a push launch of its own; not the default -- ranks sharing a "
                            "GPU (rehearsals) keep each other's pushes off the CUs, so only a run with one rank per GPU can rank the two"}
            elif inside_problems:
                out["push_inside"] = {"value": None, "preflight": {"ok": False, "problems": inside_problems, "checks": inside_checks}}
            if coll_problems:
                out["rccl_only"] = {"value": None, "preflight": {"ok": False, "problems": coll_problems},
                                    "note": "the communicator's data plane failed its pre-flight: nothing was timed on it; `value` is the "
                                            "peer-mapped plane's, which passed"}
                out["degraded"] = {"why": "the communicator's data plane failed its pre-flight", "value_is_quoted_on": plane_name(1),
                                   "problems": coll_problems}
            elif res_coll is None and world > 1:
                out["rccl_only"] = {"note": "not timed separately: " + (
                    "--no-rccl-leg" if args.no_rccl_leg else "the peer-mapped paths are off, `value` IS the communicator's data plane")}
        prob.free()
    else:
        if world != 1:
            raise SystemExit("bench: --workload irregular is a one-GPU workload (configs[4])")
        formats = {}
        best = None
        specs = [("crs", 1)] + [("scs", s) for s in args.irr_sigmas]
        # pre-flight: the stand-in at 24^3 nodes against the history the REFERENCE ITSELF produced on it (its own reader,
        # convertMatrix and solveCG on the matrix exported as .mtx; tests/golden/cg_hist_irregular_ref.json): every format
        # within north_star's 1e-12 of it, CRS and Sell-64-1 (same row order, same dot order) with identical bits
        irr_checks, irr_problems = [], []
        if not args.no_preflight:
            try:
                gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cg_hist_irregular_ref.json")))["irregular24"]
            except (OSError, ValueError, KeyError):
                gold = None
            if gold:
                ref_rr, ref_pap = np.array([float(v) for v in gold["rr"]]), np.array([float(v) for v in gold["pAp"]])
                first = None
                for fmt, sigma in specs:
                    with quiet_stdout():
                        pr = hostapi.Problem("irregular", 24, 24, 24, fmt=fmt, Cc=64, sigma=sigma)
                    cgp = hostapi.CG(pr, fused=args.fused, fuse_p=args.fuse_p, fuse_alpha=args.fuse_alpha, fuse_beta=args.fuse_beta)
                    kk = cgp.solve(gold["itermax"], 0.0)
                    rr, pap = cgp.history()
                    cgp.free()
                    pr.free()
                    label = "irregular 24^3 nodes, %s sigma %d" % (fmt, sigma)
                    dev = float(max((np.abs(rr - ref_rr) / ref_rr).max(), (np.abs(pap - ref_pap) / ref_pap).max())) if len(rr) == len(ref_rr) and len(pap) == len(ref_pap) else float("inf")
                    rec = {"case": label, "k": kk, "max_rel_deviation_from_the_reference_history": dev, "ok": kk == gold["k"] and dev <= 1e-12}
                    if sigma == 1:
                        if first is None:
                            first = (rr, pap)
                        rec["same_bits_as_crs"] = bool(np.array_equal(rr, first[0]) and np.array_equal(pap, first[1]))
                        rec["ok"] = rec["ok"] and rec["same_bits_as_crs"]
                    irr_checks.append(rec)
                    if not rec["ok"]:
                        irr_problems.append("%s: k = %d (reference %d), deviation %.3g from the reference's history (bound 1e-12)" % (label, kk, gold["k"], dev))
                if irr_problems:
                    fail_preflight(irr_checks, irr_problems, "irregular_fe_%d^3_nodes" % n)
        for fmt, sigma in specs:
            with quiet_stdout():
                prob = hostapi.Problem("irregular", n, n, n, fmt=fmt, Cc=64, sigma=sigma)
            default = prob.pack_info()["mode"]
            res = measure(prob, [default])
            d = res[default]
            name = "crs" if fmt == "crs" else "scs_C64_sigma%d" % sigma
            workload = "irregular_fe_%d^3_nodes_%s" % (n, name)
            kern = kernel_name(fmt, default, bool(L.sb_matrix_crs_kernel(prob.matrix)))
            tr = pmc_traffic(workload, kern, version)
            # (a native CRS kernel without the fused p.Ap adds a dot pass over p and Ap: 16 B/row)
            dot_pass = bool(d["phases"] and "dot_pass" in d["phases"])
            cg_moved = d["moved"] + vector_bytes(prob.nr) + (16.0 * prob.nr if dot_pass else 0.0)
            formats[name] = {
                "cg_iterations_per_s": K / d["t_clean"], "ms_per_step": 1e3 * d["t_clean"] / K,
                "fill": (prob.nnzTrue / prob.nElems) if fmt == "scs" else 1.0,
                "roofline": roofline_block(kern, d["moved"], d["alg"], d["spmv_us"], d["launches"], *tr),
                "spmv_useful_GBs": ((12.0 * prob.nnzTrue + 16.0 * prob.nr) / (d["spmv_us"] * 1e-6) / 1e9) if d["launches"] else None,
                "separate_dot_pass": dot_pass, "phases_us": phase_table(d["phases"]),
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
            "timed_repeats": repeats,
            "roofline": b["roofline"], "formats": formats, "device": L.sb_device_name().decode(),
            "preflight": ({"ok": True, "checks": irr_checks} if irr_checks else {"ok": None, "skipped": "--no-preflight or no golden"}),
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
    ap.add_argument("--all-clean", action="store_true", help="N > 1: also time the reference-layout kernel without events "
                                                             "(N = 1 always does)")
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
