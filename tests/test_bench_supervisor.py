"""bench.py's N > 1 supervisor (no GPU: the workers are a stand-in script that speaks the worker protocol).

What it must do (bench.py: supervise): relay rank 0's single JSON line; end everybody when one worker fails; return the
failing worker's exit code when the failure comes BEFORE the communicator-plane checkpoint; print the provisional line with
a "degraded" block and exit 0 when a worker dies, or hangs, BEHIND the checkpoint."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys, time
    rank = int(os.environ["RANK"]); mode = os.environ["FAKE_MODE"]; MARK = "@@sbbench "
    assert os.environ["SB_BENCH_RANK_PROCESS"] == "1" and os.environ["WORLD_SIZE"] == sys.argv[sys.argv.index("--gpus") + 1]
    print("chatter that is not protocol", flush=True)
    if mode == "die_early" and rank == 1:
        sys.stderr.write("rank 1: boom before the checkpoint\\n"); os._exit(5)
    if mode == "die_early":
        time.sleep(60)
    if rank == 0:
        print(MARK + "provisional " + json.dumps({"value": 1.0, "config": {"data_plane": "rccl data plane"}}), flush=True)
    print(MARK + "checkpoint", flush=True)
    if mode == "die_late" and rank == 1:
        sys.stderr.write("rank 1: boom behind the checkpoint\\n"); os._exit(9)
    if mode in ("die_late", "hang_late") and not (mode == "die_late" and rank == 1):
        time.sleep(60)
    if rank == 0:
        print(json.dumps({"value": 2.0, "config": {"data_plane": "peer-mapped data plane"}}), flush=True)
''')


def run(tmp_path, mode, n=3, env=None, torchrun_rank=None):
    w = tmp_path / "fake_worker.py"
    w.write_text(WORKER)
    e = dict(os.environ, SB_BENCH_WORKER_SCRIPT=str(w), FAKE_MODE=mode, SB_BENCH_LOG_DIR=str(tmp_path / "logs"), **(env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SB_BENCH_RANK_PROCESS"):
        e.pop(k, None)
    if torchrun_rank is not None:  # as a rank process of torch.distributed.run: supervises ONE worker, environment inherited
        e.update(RANK=str(torchrun_rank), LOCAL_RANK=str(torchrun_rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=e, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=120)
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    return out.returncode, lines, out.stderr.decode(), time.time() - t0


def test_relays_exactly_one_line(tmp_path):
    rc, lines, err, _ = run(tmp_path, "ok")
    assert rc == 0 and len(lines) == 1 and json.loads(lines[0])["value"] == 2.0 and "degraded" not in lines[0]
    assert "chatter that is not protocol" in err and "@@sbbench" not in err  # other stdout goes to stderr, protocol lines nowhere


def test_failure_before_the_checkpoint_is_an_error(tmp_path):
    rc, lines, err, dt = run(tmp_path, "die_early")
    assert rc == 5 and lines == [] and "rank 1 exited with code 5" in err and "boom before the checkpoint" in err and dt < 30
    assert "the workers' stderr streams are in" in err


def test_failure_behind_the_checkpoint_degrades_to_the_validated_plane(tmp_path):
    rc, lines, err, dt = run(tmp_path, "die_late")
    assert rc == 0 and len(lines) == 1 and dt < 30
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["degraded"]["value_is_quoted_on"] == "rccl data plane"
    assert "rank 1 exited with code 9" in d["degraded"]["why"] and d["degraded"]["exit_codes"]["1"] == 9
    assert any("boom behind the checkpoint" in ln for ln in d["degraded"]["stderr_tail"]["1"]) and "DEGRADED" in err
    # a degraded record is a finding, not a pass -- top-level flag, and every worker's WHOLE stderr in a file the line names
    assert d["ok"] is False
    log = d["degraded"]["stderr_files"]["1"]
    assert log.startswith(str(tmp_path)) and "boom behind the checkpoint" in open(log).read() and "exit code 9" in open(log).read()


def test_a_hang_behind_the_checkpoint_is_bounded(tmp_path):
    rc, lines, err, dt = run(tmp_path, "hang_late", env={"SB_BENCH_AFTER_CHECKPOINT_S": "2"})
    assert rc == 0 and len(lines) == 1 and dt < 30
    assert "SB_BENCH_AFTER_CHECKPOINT_S" in json.loads(lines[0])["degraded"]["why"]


def test_under_torchrun_each_rank_process_supervises_one_worker(tmp_path):
    rc, lines, _, _ = run(tmp_path, "ok", torchrun_rank=0)
    assert rc == 0 and len(lines) == 1 and json.loads(lines[0])["value"] == 2.0
    rc, lines, _, _ = run(tmp_path, "ok", torchrun_rank=2)
    assert rc == 0 and lines == []  # only rank 0 owns the line
    rc, lines, _, _ = run(tmp_path, "die_late", torchrun_rank=1)  # the rank whose worker crashed behind the checkpoint
    assert rc == 0 and lines == []
    rc, lines, _, _ = run(tmp_path, "die_early", torchrun_rank=1)
    assert rc == 5 and lines == []
