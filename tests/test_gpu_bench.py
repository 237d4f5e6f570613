"""bench.py as the driver invokes it: one JSON line, the contract's keys, every fraction <= 1, and
`--gpus N` working as typed (the parent starts the ranks itself; rehearsed here with the host transport, N ranks
sharing the one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None, want_rc=0):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, **(env or {})))
    assert out.returncode == want_rc, (out.returncode, out.stderr.decode()[-3000:])
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    if want_rc != 0 and not lines:
        return None, out.stderr.decode()
    assert len(lines) == 1, lines  # exactly ONE line on stdout
    return json.loads(lines[0]), out.stderr.decode()


def check_fractions(d):
    for k, v in d.items():
        if isinstance(v, dict):
            check_fractions(v)
        elif "frac" in k and v is not None:
            assert 0.0 <= v <= 1.0, (k, v)


def test_two_ranks_as_typed(gpu):
    """exactly the command VERDICT asks for"""
    d, err = run_bench("--gpus", "2", "--transport", "host", "--steps", "10", "--n", "32")
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["transport"].startswith("host_staged") and d["config"]["parallelism"] == "1d_block_row_x2"
    assert d["config"]["dot_allreduce"] in ("in_kernel_peer_mapped", "host_staged_gloo")
    assert d["config"]["dot_allreduce_reason"] and d["config"]["halo_exchange_reason"]  # which data plane ran, and why
    check_fractions(d)
    # self-validating: the bench bricks, on every data plane and with every SpMV kernel that was timed, against closed forms + golden + each other
    pf = d["preflight"]
    assert pf["ok"] and all(c["ok"] for c in pf["checks"]) and 1 <= len(pf["checks"]) <= 4
    assert 0 in [c["spmv_kernel_mode"] for c in pf["checks"]]  # the loop `value` is quoted on was validated
    for c in pf["checks"]:
        assert c["rr0"] == c["rr0_closed_form"] and c["pAp1"] == c["pAp1_closed_form"] and len(set(c["history_sha256_by_rank"])) == 1
        assert c["golden"] == "hpcg32_x2_scs_C64_sigma256" and c["golden_values_compared"] >= 36  # bit for bit vs the oracle's 2-rank run
    # `value` is the loop whose SpMV streams the reference's arrays; the compressed-mirror loop has a block of its own
    assert d["ok"] is True and d["config"]["spmv_kernel"] == "spmv_scs64" and d["roofline"]["kernel"] == "spmv_scs64"
    assert d["config"]["p_update_inside_spmv"] is False and 0 < d["cg_frac_of_roofline"] <= 1.0
    se = d.get("structure_exploiting")
    # self-diagnosing: per-rank step times, per-kernel breakdown, K < 100 => median of repeats, the second data plane
    assert d["timed_repeats"] == 9 and len(d["ms_per_step_repeats"]) == 9
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and pr["ms_per_step_min"] <= pr["ms_per_step_max"] <= d["ms_per_step"] * 1.0001
    assert len(pr["device"]) == 2 and len(pr["phases_us"]) == 2
    for name in ("spmv", "alpha_step", "r_update", "beta_step", "halo", "p_update"):
        assert d["phases_us"][name] > 0 and d["phases_us_max_over_ranks"][name] >= d["phases_us"][name] * 0.999, name
    if se:
        assert se["value"] > 0 and se["spmv_kernel_mode"] >= 3 and len(se["per_rank_ms_per_step"]) == 2 and se["algorithmic_speedup"] > 1
        assert ("p_update" in se["phases_us"]) != se["p_update_inside_spmv"]
    if d["config"]["dot_allreduce"] == "in_kernel_peer_mapped":
        ro = d["rccl_only"]
        assert ro["value"] > 0 and ro["dot_allreduce"] == "host_staged_gloo" and len(ro["per_rank_ms_per_step"]) == 2
        # without the peer-mapped paths: pack kernel + the local reduce of each dot in front of its all-reduce (the steps themselves
        # ride in the r / p updates), and three communicator calls per body
        assert ro["launches_per_iteration"] == 6 and ro["collective_calls_per_iteration"] == 3 and ro["phases_us"]["alpha_step"] > 0
        assert d["config"]["collective_calls_per_iteration"] == 0 and 5 <= d["config"]["launches_per_iteration"] <= 7
        if se:
            fp = 1 if se["p_update_inside_spmv"] else 0
            assert se["launches_per_iteration"] == 6 - fp and se["collective_calls_per_iteration"] == 0
            assert ro["structure_exploiting"]["value"] > 0
            pi = se["push_inside"]  # third leg: the push inside the SpMV launch, validated by its own pre-flight
            assert pi["value"] > 0 and pi["launches_per_iteration"] == 5 - fp and pi["preflight"]["ok"]
    else:
        assert "note" in d["rccl_only"]


def test_a_wrong_halo_value_fails_the_preflight_and_nothing_is_timed(gpu):
    """SB_TEST_CORRUPT_HALO=1: rank 1 sends another row's value in its first halo slot (library test hook).  The
    closed-form p.Ap of the first body and the golden history must catch it: exit code 4, no rate in the line."""
    d, err = run_bench("--gpus", "2", "--transport", "host", "--steps", "10", "--n", "32", "--no-cpu",
                       env={"SB_TEST_CORRUPT_HALO": "1"}, want_rc=4)
    assert "PRE-FLIGHT FAILED" in err and "SB_TEST_CORRUPT_HALO" in err
    assert d is not None and d["value"] is None and d["preflight"]["ok"] is False
    assert any("p.Ap of the first body" in p for p in d["preflight"]["problems"])
    assert any("golden" in p for p in d["preflight"]["problems"])
    # ... at a size without a committed golden the closed form alone must fire (and does on both bricks' sizes)
    d, err = run_bench("--gpus", "3", "--transport", "host", "--steps", "10", "--n", "16", "--no-cpu",
                       env={"SB_TEST_CORRUPT_HALO": "2"}, want_rc=4)
    assert d["value"] is None and any("closed form" in p for p in d["preflight"]["problems"])


def test_a_rank_that_dies_ends_the_run_at_once_with_its_exit_code(gpu):
    """the parent polls ALL its children; one dead rank must not leave the others in a collective for ever"""
    import time
    t0 = time.time()
    d, err = run_bench("--gpus", "3", "--transport", "host", "--steps", "10", "--n", "16", "--no-cpu",
                       env={"SB_BENCH_TEST_DIE_RANK": "2"}, want_rc=7)
    assert d is None and "rank 2 exited with code 7" in err and time.time() - t0 < 120


def _peer_mapped_up(d):
    return d["config"].get("dot_allreduce") == "in_kernel_peer_mapped" or "rccl_only" in d and "value" in d["rccl_only"]


def test_a_peer_mapped_plane_that_delivers_wrong_values_degrades_to_the_validated_plane(gpu):
    """first contact with real links must yield a line if ANY validated plane works: SB_TEST_CORRUPT_P2P_HALO=1 makes rank
    1's peer-mapped push swap two values (the communicator's plane is untouched).  The communicator's plane passes its
    pre-flight and is timed first; the peer-mapped plane fails its own: the line is quoted on the former, says so, exit 0."""
    ok, _ = run_bench("--gpus", "2", "--transport", "host", "--steps", "10", "--n", "32", "--no-cpu", "--no-push-inside-leg")
    if ok["config"]["halo_exchange"] != "peer_mapped_push_pull":
        pytest.skip("the peer-mapped halo did not come up on this box: nothing to degrade from")
    d, err = run_bench("--gpus", "2", "--transport", "host", "--steps", "10", "--n", "32", "--no-cpu",
                       env={"SB_TEST_CORRUPT_P2P_HALO": "1"})
    assert d["value"] > 0 and d["config"]["data_plane"] == "host_staged_gloo data plane"
    dg = d["degraded"]
    assert "peer-mapped data plane failed its pre-flight" in dg["why"] and dg["problems"] and "DEGRADED" in err
    pf = d["preflight"]
    assert pf["ok"] is False and pf["ok_on_the_plane_value_is_quoted_on"] is True and d["ok"] is False
    oks = [(c["case"].split(":")[0], c["ok"]) for c in pf["checks"]]  # communicator's plane (every kernel), then the peer-mapped one
    assert all(ok for pl, ok in oks if pl.startswith("host_staged")) and not any(ok for pl, ok in oks if pl.startswith("peer-mapped"))
    assert oks[0][0].startswith("host_staged") and oks[-1][0].startswith("peer-mapped")
    # (the rate is the communicator plane's: what the healthy run reports as rccl_only, within rehearsal noise)
    assert 0.3 < d["value"] / ok["rccl_only"]["value"] < 3.0


def test_a_communicator_plane_that_delivers_wrong_values_leaves_the_peer_mapped_line(gpu):
    """... and the other way round: SB_TEST_CORRUPT_HOST_EXCHANGE=1 makes rank 1's staged send / recv swap two values (the
    peer-mapped push is untouched).  Nothing is timed on the communicator's plane, no checkpoint; the peer-mapped plane passes its
    own pre-flight and carries `value`; the line says which plane failed.  Both wrong (SB_TEST_CORRUPT_HALO) stays exit code 4."""
    d, err = run_bench("--gpus", "2", "--transport", "host", "--steps", "10", "--n", "32", "--no-cpu", "--no-push-inside-leg",
                       env={"SB_TEST_CORRUPT_HOST_EXCHANGE": "1"})
    if d["config"]["halo_exchange"] != "peer_mapped_push_pull":
        pytest.skip("the peer-mapped halo did not come up on this box: `value` IS the communicator's plane (and rc would be 4)")
    assert d["value"] > 0 and d["rccl_only"]["value"] is None and d["rccl_only"]["preflight"]["problems"]
    assert d["degraded"]["value_is_quoted_on"] == "peer-mapped data plane" and "PRE-FLIGHT FAILED on the communicator's data plane" in err
    pf = d["preflight"]
    assert pf["ok"] is False and pf["ok_on_the_plane_value_is_quoted_on"] is True and d["ok"] is False
    oks = [(c["case"].split(":")[0], c["ok"]) for c in pf["checks"]]
    assert not any(ok for pl, ok in oks if pl.startswith("host_staged")) and all(ok for pl, ok in oks if pl.startswith("peer-mapped"))


def test_a_crash_in_the_peer_mapped_legs_still_yields_the_validated_line(gpu):
    d, err = run_bench("--gpus", "2", "--transport", "host", "--steps", "10", "--n", "32", "--no-cpu",
                       env={"SB_BENCH_TEST_DIE_AFTER_CHECKPOINT": "1"})
    if "degraded" not in d:
        pytest.skip("one data plane only on this box (no checkpoint): %s" % d["config"].get("dot_allreduce_reason"))
    assert d["value"] > 0 and d["preflight"]["ok"] and d["config"]["data_plane"] == "host_staged_gloo data plane"
    assert "rank 1 exited with code 9" in d["degraded"]["why"] and d["degraded"]["exit_codes"]["1"] == 9
    assert d["ok"] is False and os.path.exists(d["degraded"]["stderr_files"]["1"])


def test_a_rank_whose_wait_runs_out_takes_the_others_with_it_at_once(gpu):
    """a rank whose bounded wait runs out used to stop alone, and every peer then sat out its own 30 s in the
    next all-reduce before reporting a misleading "contribution did not arrive".  Now the failing rank poisons what it would have
    published (all-reduce slots, halo flags; kernels.hip.h: P2P_POISON) and every wait treats that as "the peer has failed".
    Library test hooks: rank 2 does not announce one halo exchange of the peer-mapped pre-flight; halo waits are bounded at
    400 ms, the all-reduce's at 60 s.  Rank 1 (waiting for rank 2's block) reports the cause, the others leave with it within
    seconds -- and, the communicator's plane having been validated and timed first, the run still yields its degraded line."""
    import time
    ok, _ = run_bench("--gpus", "3", "--transport", "host", "--steps", "10", "--n", "32", "--no-cpu", "--no-push-inside-leg")
    if ok["config"]["halo_exchange"] != "peer_mapped_push_pull":
        pytest.skip("the peer-mapped halo did not come up on this box")
    t0 = time.time()
    d, err = run_bench("--gpus", "3", "--transport", "host", "--steps", "10", "--n", "32", "--no-cpu",
                       env={"SB_TEST_DROP_PUSH_RANK": "2", "SB_TEST_DROP_PUSH_AT": "7", "SB_TEST_HALO_TIMEOUT_MS": "400",
                            "SB_P2P_TIMEOUT_MS": "60000"})
    took = time.time() - t0
    assert took < 45, took  # (without the poison: 60 s in the next all-reduce)
    assert "a neighbour's halo block did not arrive within 400 ms" in err  # the rank that saw the cause says so ...
    assert "another rank reported a communication failure" in err          # ... the others say they stopped with it
    assert "contribution to an in-kernel all-reduce did not arrive" not in err
    assert d["value"] > 0 and "degraded" in d and d["config"]["data_plane"] == "host_staged_gloo data plane"


def test_as_the_driver_launches_it_under_torch_distributed_run(gpu):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: every rank process supervises one worker"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport",
                          "host", "--steps", "10", "--warmup", "3", "--grid", "32", "--no-cpu"], stdout=subprocess.PIPE,  # ("--n" is an ambiguous prefix for torchrun's own parser)
                         stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["preflight"]["ok"] and "degraded" not in d and len(d["per_rank"]["ms_per_step"]) == 2


def test_five_ranks_next_to_the_test_process(gpu):
    """the pool allows 6 processes per GPU and this test process holds one of them: 5 ranks here; N = 6 is rehearsed from the
    command line (profiles/r03_bench_rehearsal_n6.json), where bench.py's parent stays off the GPU"""
    d, _ = run_bench("--gpus", "5", "--transport", "host", "--steps", "6", "--warmup", "2", "--n", "32", "--no-cpu")
    assert d["n_gpus"] == 5 and d["value"] > 0 and d["preflight"]["ok"]
    assert all(c["golden"] == "hpcg32_x5_scs_C64_sigma256" for c in d["preflight"]["checks"])
    assert len(d["per_rank"]["ms_per_step"]) == 5 and len(d["config"]["spmv_kernel_mode_structure_exploiting_by_rank"]) == 5


@pytest.mark.parametrize("n_ranks,p2p", [(4, "1"), (4, "0"), (3, "1")])
def test_more_ranks_and_both_data_planes(gpu, n_ranks, p2p):
    d, _ = run_bench("--gpus", str(n_ranks), "--transport", "host", "--steps", "10", "--n", "16", "--no-cpu",
                     env={"SB_P2P": p2p, "SB_P2P_HALO": p2p})
    assert d["n_gpus"] == n_ranks and d["value"] > 0
    assert len(d["config"]["spmv_kernel_mode_structure_exploiting_by_rank"]) == n_ranks  # which mirror kernel every rank's brick got
    assert d["preflight"]["ok"]
    if p2p == "0":
        assert d["config"]["dot_allreduce"] == "host_staged_gloo" and "SB_P2P=0" in d["config"]["dot_allreduce_reason"]
        assert "IS the communicator's data plane" in d["rccl_only"]["note"]


def test_single_gpu_line_has_the_contract_keys(gpu):
    d, _ = run_bench("--steps", "40", "--warmup", "5", "--n", "48", "--cpu-iters", "10")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["config"]["workload"].startswith("hpcg_27pt_48^3")
    # `value` / `roofline` are the loop whose SpMV streams the reference's arrays, on SURVEY 8d's algorithmic bytes
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"] == "spmv_scs64" and d["config"]["spmv_kernel"] == "spmv_scs64" and d["ok"] is True
    nr = 48 ** 3
    assert r["launches_timed"] == 40 and r["bytes_per_launch"] == r["algorithmic_bytes_per_launch"]
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    # section 8d: bytes of the reference's unfused op list per iteration / ms_per_step stays under the peak
    oplist = d["cg_reference_oplist_bytes_per_iteration"]
    assert oplist == 96.0 * nr + r["algorithmic_bytes_per_launch"]
    assert abs(d["cg_frac_of_roofline"] - d["value"] * oplist / 8e12) < 1e-9 and 0 < d["cg_frac_of_roofline"] <= 1.0
    assert d["phases_us"]["spmv"] > 0 and "p_update" in d["phases_us"]  # (that loop keeps its separate p update)
    se = d["structure_exploiting"]  # the compressed-mirror loop, whole, in a block of its own: faster, and not a roofline figure
    assert se["kernel"] in ("spmv_prog_fusep", "spmv_scs64_pat_masked") and se["value"] > d["value"] and se["algorithmic_speedup"] > 1.0
    assert se["moved_bytes_per_launch"] < se["algorithmic_bytes_per_launch"] and se["phases_us"]["spmv"] < d["phases_us"]["spmv"]
    rm = se["roofline_on_moved_bytes"]
    assert "frac" not in rm and 0 < rm["frac_of_hbm_peak_on_moved_bytes"] <= 1.0 and rm["bytes_per_launch"] == se["moved_bytes_per_launch"]
    assert se["sustained"]["value"] > 0 and list(d["cg_iterations_per_s_by_spmv_kernel"].values()) == [d["value"], se["value"]]
    # (the breakdown is taken with an event after every launch, ~2-3 us each: its sum brackets the clean step time from above)
    assert d["preflight"]["ok"] and d["timed_repeats"] == 9 and 1e3 * d["ms_per_step"] <= sum(d["phases_us"].values()) <= 3e3 * d["ms_per_step"]
    assert sorted(c["spmv_kernel_mode"] for c in d["preflight"]["checks"] if "bench bricks" in c["case"]) == [0, se["spmv_kernel_mode"]]
    # informational: the same clean loop over thousands of steps in one go (a K-step window is a short burst between host-side pauses)
    assert d["sustained"]["steps"] == 4800 and d["sustained"]["value"] > 0.8 * d["value"]
    cb = d["cpu_baseline"]
    assert cb and cb["value"] > 0 and cb["cores"] >= 1 and cb["nproc"] >= cb["cores"] and cb["kind"] in ("reference", "port")
    check_fractions(d)


def test_irregular_workload_line(gpu):
    d, _ = run_bench("--workload", "irregular", "--n", "16", "--steps", "30", "--warmup", "5", "--no-cpu", "--irr-sigmas", "1,512")
    assert "Flan_1565 not available" in d["config"]["workload"]
    assert set(d["formats"]) == {"crs", "scs_C64_sigma1", "scs_C64_sigma512"}
    for f in d["formats"].values():
        assert f["cg_iterations_per_s"] > 0 and 0 < f["roofline"]["frac"] <= 1.0 and 0 < f["fill"] <= 1.0
    assert d["value"] == max(f["cg_iterations_per_s"] for f in d["formats"].values())
    # validated before it was timed: every format within 1e-12 of the history the REFERENCE produced on the stand-in at 24^3 nodes
    assert d["preflight"]["ok"] and len(d["preflight"]["checks"]) == 3
    assert all(c["ok"] and c["max_rel_deviation_from_the_reference_history"] <= 1e-12 for c in d["preflight"]["checks"])
    check_fractions(d)


def test_the_benchmark_size_line_says_what_the_placement_tuner_saw(gpu):
    """at the benchmark's own size the upload chooses by measurement which device memory the streamed arrays and the loop's vectors
    occupy; the line carries what that search saw, and the kernel `value` is quoted on is the reference-layout one"""
    d, _ = run_bench("--n", "128", "--steps", "10", "--warmup", "3", "--no-cpu", "--loops", "reference", "--passes", "clean,events",
                     "--sustained-steps", "0")
    pl = d["config"]["placement"]
    assert pl and pl["probes_timed"] >= 3 and pl["us_kept"] <= pl["us_first_pair"] <= pl["us_slowest"] * 1.0001
    assert d["config"]["spmv_kernel"] == "spmv_scs64" and d["roofline"]["kernel"] == "spmv_scs64" and "structure_exploiting" not in d
    assert d["roofline"]["bytes_per_launch"] == 706234368.0 and 0.55 < d["roofline"]["frac"] <= 0.80  # SURVEY 8d bytes; HBM-bound
    # (informational, beside the data-sheet peak: what this device streams, measured in the same process)
    assert 4000 < d["roofline"]["device_stream_read_GBs"] < 8000 and 0.6 < d["roofline"]["achieved_over_device_stream_read"] < 1.05
    assert d["preflight"]["ok"] and [c["golden"] for c in d["preflight"]["checks"] if "bench bricks" in c["case"]] == ["hpcg128_x1_scs_C64_sigma256"]
    check_fractions(d)
