#!/usr/bin/env python3
"""Generate tests/golden/cg_hist_tree.json: CG residual histories of the oracle in the GPU's OWN dot order
(oracle/sb_oracle.c: dot "tree" = the fixed order of DESIGN 4.3, rank sums as a pairwise tree), i.e. the
values the HIP path must reproduce BIT FOR BIT:

  * one rank at the BASELINE sizes -- configs[1] (64^3, Sell-64-1), configs[2] (128^3, Sell-64-256; also
    Sell-64-1 and CRS) -- so the -m gpu tests assert array_equal at the benchmark size without running the
    oracle on the GPU box (tests/test_gpu_cg.py);
  * P ranks, bricks stacked in z, in the bench's default format (Sell-64-256): 32^3 per rank for
    P = 2, 3, 4, 6, 8 and BASELINE configs[3]'s own brick, 128^3 per rank, for P = 2, 4, 8 -- what
    `bench.py --gpus N` checks its pre-flight solve against before it times anything (first contact of the
    multi-rank legs with real xGMI links must be self-validating).

Under the tree order the history depends on the row order of the vectors (sigma > 1 permutes them) and on
the rank count, hence one entry per (n, P, format).  CPU only, pure oracle; ~10 minutes, ~9 GB at 128^3 x 8.
Values are written with %.17e (round-trip exact for fp64).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cg_hist_tree.json")
PREFLIGHT_ITERS = 20  # bench.py's pre-flight solve: itermax (k = 1 .. 19 loop bodies)


def key(n, P, fmt, Cc, sigma):
    return "hpcg%d_x%d_%s" % (n, P, "crs" if fmt == "crs" else "scs_C%d_sigma%d" % (Cc, sigma))


def run(n, P, fmt, Cc, sigma, itermax):
    t0 = time.time()
    locs = [po.GMatrix.generate(n, n, n, r, P) for r in range(P)]
    plans = po.Plans(locs) if P > 1 else None
    o = po.cg(locs, plans, itermax=itermax, fmt=fmt, Cc=Cc, sigma=sigma, dot="tree", rank_sum="tree")
    for g in locs:
        g.free()
    print("%-32s k=%d  rr0=%.17e  %.1fs" % (key(n, P, fmt, Cc, sigma), o["k"], o["rr"][0], time.time() - t0), flush=True)
    return {"n": n, "ranks": P, "fmt": fmt, "C": Cc, "sigma": sigma, "itermax": itermax, "k": o["k"],
            "rr": ["%.17e" % v for v in o["rr"]], "pAp": ["%.17e" % v for v in o["pAp"]]}


def main():
    out = {"_comment": "oracle CG histories in the GPU's dot order (dot=tree, rank_sum=tree): bit-for-bit targets; "
                       "made by tests/golden/make_golden_tree.py"}
    cases = [(64, 1, "scs", 64, 1, 60), (128, 1, "scs", 64, 1, 60), (128, 1, "scs", 64, 256, 60), (128, 1, "crs", 64, 1, 60),
             (32, 1, "scs", 64, 256, PREFLIGHT_ITERS)]
    cases += [(32, P, "scs", 64, 256, PREFLIGHT_ITERS) for P in (2, 3, 4, 5, 6, 7, 8)]
    cases += [(128, P, "scs", 64, 256, PREFLIGHT_ITERS) for P in (2, 4, 8)]
    # the bench's other formats (`--fmt crs`, `--sigma 1`): the small pre-flight problem on 1, 2, 4, 8 ranks
    cases += [(32, P, fmt, 64, 1, PREFLIGHT_ITERS) for fmt in ("crs", "scs") for P in (1, 2, 4, 8)]
    only = sys.argv[1:]
    if only and os.path.exists(OUT):
        out = json.load(open(OUT))
    for n, P, fmt, Cc, sigma, it in cases:
        k = key(n, P, fmt, Cc, sigma)
        if only and not any(s in k for s in only):
            continue
        out[k] = run(n, P, fmt, Cc, sigma, it)
        json.dump(out, open(OUT, "w"), indent=0)


if __name__ == "__main__":
    main()
