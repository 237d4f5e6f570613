#!/usr/bin/env python3
"""Generate tests/golden/cg_hist_exact.json: the CG residual history of HPCG 64^3 (150 iterations) and 128^3
(60 iterations) with every dot product computed in twice the working precision and rounded once
(oracle/sb_oracle.c: orc_ddot_exact, Ogita-Rump-Oishi Dot2) -- the yardstick for the tolerance story at the
BASELINE sizes: the reference's sequential ddot (src/solver.c:41-62) carries a rounding error that grows with
n, the GPU's fixed tree order one that grows with log n; both are measured against THIS history
(tests/test_oracle_pinning.py on the CPU, tests/test_gpu_cg.py on the GPU).

CPU only, pure oracle (no reference code involved): run anywhere; ~2 minutes.  Also records, for the
report, the deviation of the reference history (tests/golden/cg_hist_1rank.json, captured from the reference
itself) and of the oracle's tree-order history (bit-identical to the GPU by the -m gpu tests) from it.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ref = json.load(open(os.path.join(OUT, "cg_hist_1rank.json")))
out = {"_comment": "r.r / p.Ap of CG with exactly rounded dots (Dot2); *_dev = max relative deviation of the "
                   "reference's sequential-sum history / the tree-order (GPU) history from it while r.r/r.r0 >= 1e-20, "
                   "per iteration ('rel') and normalised by the initial residual norm ('norm')"}
for n, it in ((32, 150), (64, 150), (128, 60)):
    g = po.GMatrix.generate(n, n, n)
    e = po.cg(g, itermax=it, dot="exact")
    t = po.cg(g, itermax=it, dot="tree")
    r = np.array([float(v) for v in ref["hpcg%d" % n]["rr"]])[:len(e["rr"])]
    live = e["rr"] / e["rr"][0] >= 1e-20

    def dev(a):
        rel = float(np.max((np.abs(a - e["rr"]) / e["rr"])[live]))
        norm = float(np.max(np.abs(np.sqrt(a) - np.sqrt(e["rr"])) / np.sqrt(e["rr"][0])))
        return {"rel": rel, "norm": norm}
    out["hpcg%d" % n] = {"itermax": it, "k": e["k"], "rr": ["%.17e" % v for v in e["rr"]],
                         "pAp": ["%.17e" % v for v in e["pAp"]], "reference_dev": dev(r), "tree_dev": dev(t["rr"])}
    print(n, "reference vs exact", dev(r), "tree (GPU order) vs exact", dev(t["rr"]), flush=True)
    g.free()
json.dump(out, open(os.path.join(OUT, "cg_hist_exact.json"), "w"), indent=0)
