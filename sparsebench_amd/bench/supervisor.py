"""bench.py's N > 1 supervisor: every rank's work runs in a WORKER process under a supervisor that never touches the GPU.

Protocol on a worker's stdout (captured, never relayed): `MARK + "provisional " + json` (rank 0: a complete line quoted on the
communicator's data plane), `MARK + "checkpoint"` (every rank: that plane is validated and timed), a line starting with "{" (rank
0's final JSON line); anything else goes to stderr.  tests/test_bench_supervisor.py drives it with stand-in workers on the CPU."""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MARK = "@@sbbench "  # prefix of the worker -> supervisor lines on a worker's stdout (never relayed)
ERR_KEEP = 2 << 20   # bytes of a worker's stderr kept for the log file of a failed run


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def write_stderr_logs(ranks, procs, state):
    """every worker's whole (bounded) stderr stream into a file of its own; returns {rank: path} -- the evidence a degraded or
    failed run needs, which the 6-12 line tails in the JSON line do not hold"""
    logdir = os.environ.get("SB_BENCH_LOG_DIR", os.path.join(ROOT, "gpurun_out", "bench_logs"))
    files = {}
    try:
        os.makedirs(logdir, exist_ok=True)
        stamp = time.strftime("%Y%m%d_%H%M%S")
        for r in ranks:
            path = os.path.join(logdir, "bench_%s_rank%d.err" % (stamp, r))
            with open(path, "w") as f:
                f.write("exit code %s\n" % procs[r].returncode)
                f.writelines(state[r]["err_all"])
            files[str(r)] = path
    except OSError as e:
        sys.stderr.write("bench: could not write the workers' stderr logs: %s\n" % e)
    return files


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
        script = os.environ.get("SB_BENCH_WORKER_SCRIPT", os.path.join(ROOT, "bench.py"))  # (test hook: tests/test_bench_supervisor.py)
        procs[r] = subprocess.Popen([sys.executable, script] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        state[r] = {"checkpoint": None, "provisional": None, "final": None, "err": [], "err_all": [], "err_bytes": 0}

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
            if state[r]["err_bytes"] < ERR_KEEP:  # the whole stream (bounded) for the log file of a failed run
                state[r]["err_all"].append(ln)
                state[r]["err_bytes"] += len(ln)

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
                # a degraded record is a finding to root-cause, not a pass: top-level flag + every worker's whole stderr on disk
                line["ok"] = False
                line["degraded"] = {"why": why + " in the legs behind the communicator-plane checkpoint",
                                    "value_is_quoted_on": line["config"].get("data_plane"),
                                    "exit_codes": {str(r): procs[r].returncode for r in ranks}, "stderr_tail": tails,
                                    "stderr_files": write_stderr_logs(ranks, procs, state)}
                print(json.dumps(line), flush=True)
            else:
                sys.stderr.write("bench: checkpoint without a provisional line\n")
                return rc
        sys.stderr.write("bench: DEGRADED completion (%s): the line is quoted on the communicator's data plane\n" % why)
        return 0
    if mine0 and state[0]["final"]:  # a failure line (pre-flight): relay it, keep the exit code
        print(state[0]["final"], flush=True)
    files = write_stderr_logs(ranks, procs, state)
    if files:
        sys.stderr.write("bench: the workers' stderr streams are in %s\n" % ", ".join(sorted(files.values())))
    return rc


