#!/usr/bin/env python3
"""fusep_check.py -- sweep of grid shapes / formats / sigma: where does the loop take the p update inside the SpMV
(spmv_prog_fusep), and is the result bit-identical to the oracle there (history, x) -- ragged last tiles, lines that are no
multiple of the chunk height, tiny grids, CPT 4 and 8, mapped and simple windows, the CRS mirror."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from sparsebench_amd import capi, hostapi  # noqa: E402

L = capi.init(0)
shapes = [(128, 5, 3), (128, 3, 1), (256, 3, 2), (128, 128, 2), (192, 4, 4), (130, 6, 5), (200, 7, 3), (64, 64, 3), (128, 1, 1), (512, 2, 1),
          (128, 9, 7), (256, 256, 1), (384, 5, 2), (129, 4, 4), (127, 8, 8), (96, 96, 4), (128, 16, 16), (16, 16, 16), (100, 100, 3)]
bad = 0
fused = 0
for dims in shapes:
    g = po.GMatrix.generate(*dims)
    for fmt, sigma in (("scs", 1), ("scs", 256), ("scs", 100000), ("crs", 1)):
        o = po.cg(g, itermax=25, fmt=fmt, Cc=64, sigma=sigma, dot="tree", want_x=True)
        prob = hostapi.Problem("generate", *dims, fmt=fmt, Cc=64, sigma=sigma)
        cg = hostapi.CG(prob, fuse_p=1)
        fp = cg.fuse_p()
        k = cg.solve(25, 0.0)
        rr, pap = cg.history()
        ok = k == o["k"] and np.array_equal(rr, o["rr"]) and np.array_equal(pap, o["pAp"]) and np.array_equal(cg.solution(), o["x"][0])
        fused += fp
        bad += not ok
        print("%-16s %s sigma %-6d rows %-7d mode %d fused-p %d launches %d  %s" % (dims, fmt, sigma, g.nr, prob.pack_info()["mode"], fp, cg.launches_per_body(),
                                                                                   "bit-identical" if ok else "*** MISMATCH ***"), flush=True)
        cg.free()
        prob.free()
    g.free()
print("cases with the p update inside the SpMV: %d; mismatches: %d" % (fused, bad))
sys.exit(1 if bad else 0)
