"""The oracle is only trusted after it reproduces (a) the reference's own golden files,
(b) histories captured from the reference compiled in place, (c) that build live, when
oracle/_ref is present.  CPU only."""
import os
import re

import numpy as np
import pytest

from conftest import REFDATA, load_json
from oracle import pyoracle as po


def parse_in(path):
    """tests/data/expected/*.in of the reference: `m->field = v` lines and `name: a, b, ` lists"""
    d = {}
    for line in open(path):
        m = re.match(r"m->(\w+) = (\d+)", line)
        if m:
            d[m.group(1)] = int(m.group(2))
            continue
        m = re.match(r"(\w+): (.*)", line)
        if m:
            vals = [v for v in m.group(2).replace(" ", "").split(",") if v]
            d[m.group(1)] = np.array([float(v) for v in vals])
    return d


@pytest.mark.parametrize("name", ["test0", "test8"])
@pytest.mark.parametrize("C", [1, 2, 4])
def test_scs_layout_matches_reference_fixture(name, C):
    exp = parse_in(os.path.join(REFDATA, "%s_C_%d_sigma_1.in" % (name, C)))
    g = po.GMatrix.from_mtx(os.path.join(REFDATA, name + ".mtx"))
    s = g.to_scs(C, 1)
    for f in ("nr", "nc", "nnz", "C", "sigma", "nChunks", "nrPadded", "nElems"):
        assert getattr(s, f) == exp[f], f
    assert g.totalNr == exp["totalNr"] and g.totalNnz == exp["totalNnz"]
    for f in ("oldToNewPerm", "newToOldPerm", "chunkLens", "chunkPtr", "colInd", "val"):
        assert np.array_equal(np.asarray(getattr(s, f), dtype=np.float64), exp[f]), f


def test_spmv_matches_reference_fixture():
    txt = open(os.path.join(REFDATA, "test0_spmv_x_1.in")).read()
    exp = np.array([float(v) for v in txt.split("=")[1].replace(" ", "").split(",") if v])
    g = po.GMatrix.from_mtx(os.path.join(REFDATA, "test0.mtx"))
    assert np.array_equal(g.spmv(np.ones(g.nc)), exp)
    for C in (1, 2, 4):
        assert np.array_equal(g.to_scs(C, 1).spmv(np.ones(g.nc)), exp)


def test_spmv_matches_captured_reference_outputs():
    ref = load_json("spmv_ref.json")
    for nm, y in ref.items():
        g = po.GMatrix.from_mtx(os.path.join(REFDATA, nm + ".mtx"))
        assert np.array_equal(g.spmv(np.ones(g.nc)), np.array([float(v) for v in y])), nm


def test_scs_layout_matches_patched_reference_incl_sigma():
    ref = load_json("scs_layout_fix.json")
    for key, d in ref.items():
        nm, C, sg = re.match(r"(test\d+)_C(\d+)_s(\d+)", key).groups()
        g = po.GMatrix.from_mtx(os.path.join(REFDATA, nm + ".mtx"))
        s = g.to_scs(int(C), int(sg))
        for f in ("nChunks", "nrPadded", "nElems"):
            assert getattr(s, f) == d[f], (key, f)
        for f in ("chunkPtr", "chunkLens", "colInd", "oldToNewPerm", "newToOldPerm"):
            assert np.array_equal(getattr(s, f), np.array(d[f], dtype=np.uint32)), (key, f)
        assert np.array_equal(s.val, np.array([float(v) for v in d["val"]])), key
        x = np.arange(1, g.nc + 1, dtype=np.float64)
        ylit = np.array([float(v) for v in d["y_literal_x_iota"]])
        assert np.array_equal(s.spmv_literal(x), ylit), key
        # fixed semantics == CRS result in original order, for every sigma
        assert np.array_equal(s.spmv(x), g.spmv(x)), key


CASES = [("band_klein", None), ("hpcg8", 8), ("hpcg16", 16), ("hpcg32", 32)]


@pytest.mark.parametrize("name,n", CASES)
def test_cg_history_bit_identical_to_reference(golden_1rank, name, n):
    gd = golden_1rank[name]
    g = (po.GMatrix.generate(n, n, n) if n else
         po.GMatrix.from_mtx(os.path.join(REFDATA, "matrix_band_klein.mtx")))
    o = po.cg(g, itermax=gd["itermax"], dot="seq")
    assert o["k"] == gd["k"]
    assert np.array_equal(o["rr"], np.array([float(v) for v in gd["rr"]]))
    assert np.array_equal(o["pAp"], np.array([float(v) for v in gd["pAp"]]))
    # SCS (incl. sigma > 1) keeps the per-row order, so the history is the CRS one
    s = po.cg(g, itermax=gd["itermax"], fmt="scs", Cc=64, sigma=256, dot="seq")
    assert np.array_equal(s["rr"], o["rr"]) and np.array_equal(s["pAp"], o["pAp"])


def test_cg_history_64_bit_identical_to_reference(golden_1rank):
    gd = golden_1rank["hpcg64"]
    o = po.cg(po.GMatrix.generate(64, 64, 64), itermax=60, dot="seq")
    assert np.array_equal(o["rr"], np.array([float(v) for v in gd["rr"]])[:len(o["rr"])])
    assert np.array_equal(o["pAp"], np.array([float(v) for v in gd["pAp"]])[:len(o["pAp"])])


def test_known_answers_of_baseline_md(golden_1rank):
    """BASELINE.md section 3 / SURVEY 8c: closed form r.r0 and the quoted values"""
    for n in (8, 16, 32, 64, 128):
        rr0 = float(golden_1rank["hpcg%d" % n]["rr"][0])
        m = n - 2
        assert rr0 == m ** 3 + 600 * m ** 2 + 3072 * m + 3200
    assert float(golden_1rank["hpcg8"]["rr"][1]) == 1.07735461628460198e4
    assert float(golden_1rank["hpcg64"]["rr"][1]) == 1.12311444588247687e6
    assert float(golden_1rank["hpcg128"]["rr"][1]) == 6.36136639649158530e6
    assert golden_1rank["band_klein"]["k"] == 3
    assert [float(v) for v in golden_1rank["band_klein"]["rr"]] == [100.0, 0.0]


@pytest.mark.parametrize("key,P,n", [("hpcg16_x2", 2, 16), ("hpcg16_x4", 4, 16), ("hpcg8_x8", 8, 8)])
def test_multirank_history_matches_mpi_reference(golden_mpi, key, P, n):
    """P ranks emulated in one process vs the reference under mpiexec -n P (MPICH).
    The all-reduce order of MPI is implementation defined; recursive doubling ==
    pairwise tree reproduces it bit for bit here."""
    gd = golden_mpi[key]
    locs = [po.GMatrix.generate(n, n, n, r, P) for r in range(P)]
    plans = po.Plans(locs)
    o = po.cg(locs, plans, itermax=gd["itermax"], dot="seq", rank_sum="tree")
    rr = np.array([float(v) for v in gd["rr"]])
    pap = np.array([float(v) for v in gd["pAp"]])
    assert len(o["rr"]) == len(rr) and len(o["pAp"]) == len(pap)
    assert np.array_equal(o["rr"], rr)
    assert np.array_equal(o["pAp"], pap)


def test_multirank_band_klein_matches_mpi_reference(golden_mpi):
    gd = golden_mpi["band_klein_x2"]
    path = os.path.join(REFDATA, "matrix_band_klein.mtx")
    locs = [po.GMatrix.from_mtx(path, r, 2) for r in range(2)]
    plans = po.Plans(locs)
    o = po.cg(locs, plans, itermax=150, dot="seq", rank_sum="tree")
    assert np.array_equal(o["rr"], np.array([float(v) for v in gd["rr"]]))


@pytest.mark.skipif(not po.ref_available("crs"), reason="oracle/_ref not built here")
def test_live_reference_build_agrees():
    ref = po.Ref("crs")
    ref.setup("generate", 12, 10, 9)
    g = po.GMatrix.generate(12, 10, 9)
    rp, col, val = ref.csr()
    assert np.array_equal(rp, g.rowPtr) and np.array_equal(col, g.col) and np.array_equal(val, g.val)
    rng = np.random.default_rng(7)
    x = rng.standard_normal(g.nc)
    y = rng.standard_normal(g.nr)
    assert np.array_equal(ref.spmv(x), g.spmv(x))
    assert np.array_equal(ref.waxpby(1.0, y, -0.37, x), po.waxpby(1.0, y, -0.37, x))
    assert np.array_equal(ref.waxpby(2.5, y, 1.0, x), po.waxpby(2.5, y, 1.0, x))
    assert np.array_equal(ref.waxpby(2.5, y, -3.0, x), po.waxpby(2.5, y, -3.0, x))
    assert ref.ddot(x, y) == po.ddot_seq(x, y)
    h = ref.solve_cg(40)
    o = po.cg(g, itermax=40)
    assert h["k"] == o["k"] and np.array_equal(h["rr"], o["rr"]) and np.array_equal(h["pAp"], o["pAp"])


def test_tree_dot_is_within_the_sequential_sums_own_error():
    """The fixed-order (tree) dot of the HIP kernels differs from the reference's
    sequential sum by no more than that sum's a-priori rounding bound (n-1)*u*sum|t|,
    and is the closer of the two to the exactly rounded result."""
    import math
    rng = np.random.default_rng(3)
    for n in (1, 63, 64, 65, 1000, 4097, 262144):
        x = rng.standard_normal(n)
        y = rng.standard_normal(n)
        exact = math.fsum((x * y).tolist())
        seq, tree = po.ddot_seq(x, y), po.ddot_tree(x, y)
        bound = (n - 1) * 2.0 ** -53 * float(np.sum(np.abs(x * y))) + 1e-300
        assert abs(seq - tree) <= bound
        assert abs(tree - exact) <= abs(seq - exact) + 8 * 2.0 ** -53 * abs(exact)
        q = po.ddot_partials(x, y)
        assert po.reduce_final(q) == tree


def f(a):
    return np.array([float(v) for v in a])


@pytest.mark.parametrize("n,its", [(32, 150), (64, 150), (128, 25)])
def test_cg_tolerance_story_against_exactly_rounded_dots(n, its, golden_1rank):
    """CG-level evidence at the BASELINE sizes (not only a dot-level argument): tests/golden/cg_hist_exact.json
    is the history with every dot computed in twice the working precision and rounded once
    (orc_ddot_exact; script tests/golden/make_golden_exact.py).  Against it
      - the tree-order history (what the GPU produces bit for bit, tests/test_gpu_cg.py) stays within 1e-12
        per iteration -- north_star's tolerance -- in fact within 1e-13;
      - the reference's own sequential-sum history (captured from the reference, cg_hist_1rank.json) is off by
        8e-13 (32^3), 2.5e-11 (64^3), 3.2e-10 (128^3): the reference is the outlier, by 10x to 10^4x."""
    ex = load_json("cg_hist_exact.json")["hpcg%d" % n]
    e_rr, e_pap = f(ex["rr"])[:its - 1], f(ex["pAp"])[:its - 1]
    g = po.GMatrix.generate(n, n, n)
    e = po.cg(g, itermax=its, dot="exact")
    assert np.array_equal(e["rr"], e_rr) and np.array_equal(e["pAp"], e_pap)  # the committed fixture is reproducible
    t = po.cg(g, itermax=its, dot="tree")
    ref = f(golden_1rank["hpcg%d" % n]["rr"])[:its - 1]
    live = e_rr / e_rr[0] >= 1e-20
    d_tree = (np.abs(t["rr"] - e_rr) / e_rr)[live].max()
    d_ref = (np.abs(ref - e_rr) / e_rr)[live].max()
    assert d_tree <= 1e-12 and d_tree <= 1e-13
    assert d_ref > 5 * d_tree
    if its == ex["itermax"]:  # the numbers quoted in DESIGN.md
        assert abs(d_ref - ex["reference_dev"]["rel"]) <= 1e-3 * d_ref and abs(d_tree - ex["tree_dev"]["rel"]) <= 1e-3 * d_tree
    g.free()


# ---- the irregular-nnz input class (BASELINE configs[4]'s stand-in) pinned on the reference itself -------------------
def _irregular_oracle_matrix(n):
    import hashlib
    from sparsebench_amd import hostapi
    p = hostapi.Problem("irregular", n, n, n, fmt="crs", upload=False)
    rp = p.array("rowPtr").copy()
    col, val = p.gm_entries()
    g = po.GMatrix.from_csr(rp, col, val, nc=p.nc)
    p.free()
    h = hashlib.sha256()
    for a in (rp.astype(np.uint32), col.astype(np.uint32), val.astype(np.float64)):
        h.update(np.ascontiguousarray(a).tobytes())
    return g, h.hexdigest()


@pytest.mark.parametrize("n", [12, 24, 80])
def test_irregular_stand_in_history_bit_identical_to_the_reference(n):
    """tests/golden/cg_hist_irregular_ref.json: the reference's own reader + convertMatrix + solveCG run on the stand-in
    exported as .mtx (make_golden_irregular_ref.py; the reference built exactly the generator's CRS arrays from the
    file -- fingerprint below).  The oracle with the reference's sequential dot reproduces every r.r / p.Ap bit for
    bit: the oracle is pinned on long rows, far couplings and 2 M distinct values too, not only on stencils.  And the
    GPU's dot order (oracle dot='tree', bit-identical to the HIP path by tests/test_gpu_irregular.py) stays within
    north_star's 1e-12 of the REFERENCE ITSELF on this input (observed 2e-14 / 9e-14; at the FULL size of the bench's irregular
    workload -- n = 80: 1 536 000 rows, 94 385 718 nonzeros, VERDICT r3 item 5 -- 8.8e-13 / 9.7e-13, where the tree order is
    3.4e-15 from a CG with exactly rounded dots and the reference's sequential sum 8.8e-13: ~40 s and ~4 GB on the CPU)."""
    e = load_json("cg_hist_irregular_ref.json")["irregular%d" % n]
    rr = np.array([float(v) for v in e["rr"]])
    pap = np.array([float(v) for v in e["pAp"]])
    g, sha = _irregular_oracle_matrix(n)
    assert sha == e["matrix_sha256"] and g.nr == e["rows"] and g.nnzTrue == e["nnz"]
    o = po.cg(g, itermax=e["itermax"], dot="seq")
    assert o["k"] == e["k"] and np.array_equal(o["rr"], rr) and np.array_equal(o["pAp"], pap)
    t = po.cg(g, itermax=e["itermax"], dot="tree")
    assert (np.abs(t["rr"] - rr) / rr).max() <= 1e-12 and (np.abs(t["pAp"] - pap) / pap).max() <= 1e-12
    g.free()


@pytest.mark.skipif(not po.ref_available("crs"), reason="oracle/_ref not built (no /root/reference here)")
def test_live_reference_on_the_irregular_stand_in(tmp_path):
    """the same link live: export 12^3 as .mtx now, let the reference read and solve it, compare with the committed history"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(REFDATA), "make_golden_irregular_ref.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    nr, nc, rp, col, val = mk.stand_in(12)
    path = str(tmp_path / "irregular_12.mtx")
    mk.write_mtx(path, nr, rp, col, val)
    ref = po.Ref("crs")
    ref.setup(path)
    rrp, rcol, rval = ref.csr()
    assert np.array_equal(rrp, rp) and np.array_equal(rcol, col) and np.array_equal(rval, val)
    h = ref.solve_cg(40)
    e = load_json("cg_hist_irregular_ref.json")["irregular12"]
    assert np.array_equal(h["rr"], np.array([float(v) for v in e["rr"]])) and h["k"] == e["k"]
