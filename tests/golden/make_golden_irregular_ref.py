#!/usr/bin/env python3
"""Generate tests/golden/cg_hist_irregular_ref.json: the REFERENCE ITSELF (oracle/_ref/libsbref_crs.so, the reference's
own sources compiled in place by oracle/build_ref.sh) run on the irregular-nnz stand-in of BASELINE configs[4].

SuiteSparse Flan_1565 cannot be fetched, so configs[4] runs on the committed generator host/sbh_irregular.c -- an input
class (rows of 3..99 entries, far couplings, ~2 M distinct values of either sign) the reference had never seen and the
oracle was therefore not pinned on.  This script closes that link in the build container: the stand-in at 12^3, 24^3 and -- VERDICT r3 item 5 -- at the FULL size of bench.py's irregular workload (80^3 nodes: 1 536 000 rows,
94 385 718 nonzeros)
is written out as a Matrix Market file (general, 1-based, %.17g: exact), read by the reference's own reader
(src/matrix.c:123-269), converted and solved by its own solveCG (src/CGSolver.c:62-141), and every r.r / p.Ap it computes
is captured at full precision (ddot wrapped at link time, oracle/ref_shim.c).  The .mtx files are scratch (60 MB; 3 GB at full size); the
histories plus a fingerprint of the matrix are committed.  tests/test_oracle_pinning.py then asserts
oracle(sequential dot) on the generator's matrix == these histories bit for bit, and the -m gpu tests compare the HIP
path with them within the documented bound.

Data only: no reference source text is stored.  Build container only (needs /root/reference via oracle/_ref).
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from sparsebench_amd import hostapi  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cg_hist_irregular_ref.json")
ITERMAX = 40


def stand_in(n):
    p = hostapi.Problem("irregular", n, n, n, fmt="crs", upload=False)
    rp = p.array("rowPtr").copy()
    col, val = p.gm_entries()
    nr, nc = p.nr, p.nc
    p.free()
    return nr, nc, rp, col, val


def fingerprint(rp, col, val):
    h = hashlib.sha256()
    for a in (rp.astype(np.uint32), col.astype(np.uint32), val.astype(np.float64)):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def write_mtx(path, nr, rp, col, val, chunk=1 << 21):
    """general, 1-based, %.17g (exact); written in chunks so that the full-size stand-in (94 M entries, a 3 GB file) does not
    need 94 M Python objects at once"""
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("%d %d %d\n" % (nr, nr, len(col)))
        rows = np.repeat(np.arange(1, nr + 1, dtype=np.int64), np.diff(rp.astype(np.int64)))
        for a in range(0, len(col), chunk):
            b = min(len(col), a + chunk)
            f.write("".join("%d %d %.17g\n" % t for t in zip(rows[a:b].tolist(), (col[a:b].astype(np.int64) + 1).tolist(), val[a:b].tolist())))


def main():
    """default: regenerate every entry (12^3, 24^3 and the full-size 80^3 nodes of bench.py --workload irregular; the last one
    writes a 3 GB scratch file and takes the reference's reader ~10 min and ~10 GB).  `--small`: 12^3 and 24^3 only, keeping a
    committed full-size entry."""
    small = "--small" in sys.argv
    if not po.ref_available("crs"):
        raise SystemExit("oracle/_ref/libsbref_crs.so is missing: run oracle/build_ref.sh in the build container")
    out = {"_comment": "r.r / p.Ap of the reference's own solveCG (strict-IEEE CRS build, 1 rank) on the irregular stand-in "
                       "exported as .mtx; matrix_sha256 = sha256(rowPtr u32 | col u32 | val f64) of the CRS arrays the "
                       "reference built from the file; made by tests/golden/make_golden_irregular_ref.py"}
    if small and os.path.exists(OUT):
        old = json.load(open(OUT))
        if "irregular80" in old:
            out["irregular80"] = old["irregular80"]
    ref = po.Ref("crs")
    for n in (12, 24) if small else (12, 24, 80):
        nr, nc, rp, col, val = stand_in(n)
        with tempfile.TemporaryDirectory(dir=os.environ.get("SB_SCRATCH", None)) as d:
            path = os.path.join(d, "irregular_%d.mtx" % n)
            write_mtx(path, nr, rp, col, val)
            ref.setup(path)
            rrp, rcol, rval = ref.csr()
            h = ref.solve_cg(ITERMAX)
        same = np.array_equal(rrp, rp) and np.array_equal(rcol, col) and np.array_equal(rval, val)
        print("irregular %d^3: %d rows, %d nonzeros; reference's CRS arrays == generator's: %s; k = %d" % (n, nr, len(col), same, h["k"]),
              flush=True)
        if not same:
            raise SystemExit("the reference built a different matrix from the file than the generator's")
        out["irregular%d" % n] = {"nodes_per_edge": n, "rows": int(nr), "nnz": int(len(col)), "itermax": ITERMAX, "k": int(h["k"]),
                                  "matrix_sha256": fingerprint(rrp, rcol, rval),
                                  "rr": ["%.17e" % v for v in h["rr"]], "pAp": ["%.17e" % v for v in h["pAp"]]}
    json.dump(out, open(OUT, "w"), indent=0)


if __name__ == "__main__":
    main()
