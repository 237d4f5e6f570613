"""Closed-form known answers of the HPCG generator + solveCG (sparsebench_amd/knownanswers.py), the committed
tree-order goldens (tests/golden/cg_hist_tree.json) and bench.py's pre-flight gate, all on the CPU.

Anchors: BASELINE.md section 3 (values the reference prints / the survey probed with the reference itself:
r.r of the prologue and p.Ap of the first body for 8^3 ... 128^3 and 16^3 x 4 ranks) and the oracle's P-rank runs."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from sparsebench_amd import knownanswers as ka  # noqa: E402
from sparsebench_amd import srchash  # noqa: E402

# r.r0 / p.Ap1 captured from the reference (strict-IEEE build)
REFERENCE_VALUES = {
    (8, 8, 8): (4.3448e4, 7.63104e5),
    (16, 16, 16): (1.66552e5, 2.857568e6),
    (32, 32, 32): (6.6236e5, 1.1049312e7),
    (64, 64, 64): (2.738392e6, 4.353008e7),
    (128, 128, 128): (1.1916248e7, 1.73568864e8),
    (16, 16, 64): (4.93912e5, 8.37872e6),  # 16^3 x 4 ranks under mpiexec
}


@pytest.mark.parametrize("dims", sorted(REFERENCE_VALUES))
def test_closed_forms_reproduce_the_reference_values(dims):
    rr0, pap1 = REFERENCE_VALUES[dims]
    assert ka.hpcg_rr0(*dims) == rr0 and ka.hpcg_pAp1(*dims) == pap1
    n = dims[0]
    if dims[0] == dims[1] == dims[2]:  # BASELINE.md's cube form
        assert ka.hpcg_rr0(*dims) == (n - 2) ** 3 + 600 * (n - 2) ** 2 + 3072 * (n - 2) + 3200
        assert ka.hpcg_nnz(*dims) == (3 * n - 2) ** 3


@pytest.mark.parametrize("nx,ny,nz,P", [(4, 5, 3, 1), (6, 4, 2, 3), (8, 8, 8, 2), (5, 7, 4, 5), (2, 2, 2, 1), (3, 1, 2, 2)])
def test_closed_forms_equal_the_oracle_on_odd_shapes_and_rank_counts(nx, ny, nz, P):
    """non-cubic bricks, P ranks stacked in z, both of the oracle's dot orders (integers: every order is exact)"""
    locs = [po.GMatrix.generate(nx, ny, nz, r, P) for r in range(P)]
    assert sum(g.nnzTrue for g in locs) == ka.hpcg_nnz(nx, ny, nz * P)
    plans = po.Plans(locs) if P > 1 else None
    for dot in ("seq", "tree"):
        o = po.cg(locs, plans, itermax=4, dot=dot, rank_sum="tree")
        assert o["rr"][0] == ka.hpcg_rr0(nx, ny, nz * P), (dot, o["rr"][0])
        assert o["pAp"][0] == ka.hpcg_pAp1(nx, ny, nz * P), (dot, o["pAp"][0])
    for g in locs:
        g.free()


def _goldens():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "cg_hist_tree.json")))


def test_tree_goldens_start_with_the_closed_forms_and_match_a_fresh_oracle_run():
    gold = _goldens()
    cases = [k for k in gold if not k.startswith("_")]
    assert {"hpcg64_x1_scs_C64_sigma1", "hpcg128_x1_scs_C64_sigma1", "hpcg128_x1_scs_C64_sigma256", "hpcg128_x8_scs_C64_sigma256",
            "hpcg32_x2_scs_C64_sigma256", "hpcg32_x6_scs_C64_sigma256", "hpcg32_x8_scs_C64_sigma256"} <= set(cases)
    for k in cases:
        g = gold[k]
        assert float(g["rr"][0]) == ka.hpcg_rr0(g["n"], g["n"], g["n"] * g["ranks"]), k
        assert float(g["pAp"][0]) == ka.hpcg_pAp1(g["n"], g["n"], g["n"] * g["ranks"]), k
    for k in ("hpcg32_x3_scs_C64_sigma256", "hpcg32_x8_scs_C64_sigma256", "hpcg64_x1_scs_C64_sigma1"):  # regenerate the cheap ones
        g = gold[k]
        locs = [po.GMatrix.generate(g["n"], g["n"], g["n"], r, g["ranks"]) for r in range(g["ranks"])]
        plans = po.Plans(locs) if g["ranks"] > 1 else None
        o = po.cg(locs, plans, itermax=g["itermax"], fmt=g["fmt"], Cc=g["C"], sigma=g["sigma"], dot="tree", rank_sum="tree")
        assert np.array_equal(o["rr"], np.array([float(v) for v in g["rr"]])), k
        assert np.array_equal(o["pAp"], np.array([float(v) for v in g["pAp"]])), k
        for m in locs:
            m.free()


def test_the_benchs_preflight_gate_accepts_the_golden_and_rejects_a_wrong_halo_value():
    from sparsebench_amd.bench import preflight as bench
    gold = _goldens()
    key = bench.golden_key(32, 4, "scs", 64, 256)
    g = gold[key]
    rr, pap = np.array([float(v) for v in g["rr"]]), np.array([float(v) for v in g["pAp"]])
    rec, bad = bench.check_history("t", rr, pap, 32, 4, key, gold)
    assert not bad and rec["golden"] == key and rec["golden_values_compared"] == len(rr) + len(pap)
    # a halo entry that carries another row's b: the first product that uses the halo (p.Ap of body 1) moves off its closed form
    locs = [po.GMatrix.generate(32, 32, 32, r, 4) for r in range(4)]
    plans = po.Plans(locs)
    es = np.ctypeslib.as_array(plans.ptr[1].elementsToSend, shape=(plans.ptr[1].totalSendCount,))
    es[0] = es[len(es) // 2 + 1]  # what SB_TEST_CORRUPT_HALO=1 does on the device (sbhip_comm.inc.h)
    o = po.cg(locs, plans, itermax=20, fmt="scs", Cc=64, sigma=256, dot="tree", rank_sum="tree")
    rec, bad = bench.check_history("t", o["rr"], o["pAp"], 32, 4, key, gold)
    assert any("closed form" in b and "p.Ap" in b for b in bad) and any("golden" in b for b in bad), bad
    assert o["rr"][0] == rr[0]  # the prologue does not use the halo: r.r0 alone would not have caught it
    # ... and without a golden for the size, the closed form alone still fires
    rec, bad = bench.check_history("t", o["rr"], o["pAp"], 32, 4, "no such key", gold)
    assert len(bad) == 1 and "closed form" in bad[0]
    for m in locs:
        m.free()


def test_pmc_traffic_is_keyed_on_the_kernel_source_hash(tmp_path, monkeypatch):
    from sparsebench_amd.bench import line as bench
    h = srchash.csrc_hash()
    assert len(h) == 16 and h == srchash.csrc_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    doc = {"source_hash": h, "w": {"k": {"bytes_per_launch": 123.0, "source_hash": h}, "stale": {"bytes_per_launch": 5.0, "source_hash": "0" * 16}}}
    (prof / "r99_pmc_traffic.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("w", "k", "any version string")[0] == 123.0
    got = bench.pmc_traffic("w", "stale", "any version string")
    assert got[0] is None and "kernel source" in got[2]  # never a stale constant
    assert bench.pmc_traffic("w", "absent", "v")[0] is None
    # the hash really follows the sources
    monkeypatch.setattr(srchash, "CSRC", str(tmp_path))
    (tmp_path / "a.hip").write_text("x")
    h1 = srchash.csrc_hash()
    (tmp_path / "a.hip").write_text("y")
    assert srchash.csrc_hash() != h1


def test_the_bench_lines_roofline_block_is_the_survey_8d_figure():
    """`roofline.achieved` = SURVEY 8d's algorithmic bytes / the kernel's time; the structure-exploiting block is
    the only place where moved bytes are divided, and it says so and carries no key named `frac`"""
    from sparsebench_amd.bench import line
    alg = 706234368.0  # Sell-64-1 at HPCG 128^3
    r = line.roofline_block("spmv_scs64", alg, alg, 118.97, 20, 708425670.0, "profiles/r04_pmc_traffic.json", None)
    assert r["kernel"] == "spmv_scs64" and r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["achieved"] - alg / 118.97e-6 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12 and 0.74 < r["frac"] < 0.75
    assert abs(r["traffic_over_bytes"] - 1.0031) < 1e-3 and "algorithmic" in r["bytes_are"]
    m = line.roofline_block("spmv_prog_fusep", 123272448.0, 790120448.0, 28.6, 20, None, None, "x", on_moved_bytes=True)
    assert "moved" in m["bytes_are"] and m["bytes_per_launch"] < m["algorithmic_bytes_per_launch"]
    # CG iteration of the reference's unfused op list: 96 B/row + the SpMV's bytes -> 8 815 it/s at 8 TB/s
    nr = 128 ** 3
    assert 96.0 * nr + alg == 907560960.0 and abs(8e12 / (96.0 * nr + alg) - 8814.8) < 0.1
    assert line.vector_bytes(nr) == 64.0 * nr + 16.0 * nr / 256.0 and line.kernel_name("crs", 0) == "spmv_crs_split" and line.kernel_name("scs", 0) == "spmv_scs64"
