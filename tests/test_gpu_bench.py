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


def run_bench(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
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


@pytest.mark.parametrize("n_ranks,p2p", [(4, "1"), (4, "0"), (3, "1")])
def test_more_ranks_and_both_data_planes(gpu, n_ranks, p2p):
    d, _ = run_bench("--gpus", str(n_ranks), "--transport", "host", "--steps", "10", "--n", "16", "--no-cpu",
                     env={"SB_P2P": p2p, "SB_P2P_HALO": p2p})
    assert d["n_gpus"] == n_ranks and d["value"] > 0
    assert len(d["config"]["spmv_kernel_mode_by_rank"]) == n_ranks  # which SpMV kernel every rank's brick got
    if p2p == "0":
        assert d["config"]["dot_allreduce"] == "host_staged_gloo" and "SB_P2P=0" in d["config"]["dot_allreduce_reason"]


def test_single_gpu_line_has_the_contract_keys(gpu):
    d, _ = run_bench("--steps", "40", "--warmup", "5", "--n", "48", "--cpu-iters", "10")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["config"]["workload"].startswith("hpcg_27pt_48^3")
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches_timed"] == 40 and r["bytes_per_launch"] <= r["algorithmic_bytes_per_launch"] * 1.02
    assert abs(d["algorithmic_speedup"] - r["algorithmic_bytes_per_launch"] / r["bytes_per_launch"]) < 1e-9
    if r["kernel"] != "spmv_scs64":
        assert d["roofline_reference_layout"]["kernel"] == "spmv_scs64"
        assert d["roofline_reference_layout"]["bytes_per_launch"] == r["algorithmic_bytes_per_launch"]
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
    check_fractions(d)
