#!/usr/bin/env python3
"""Does a SECOND upload in the same process find a fast pair where the first did not (a device where fast pairs are rare)?  Four
uploads of the benchmark matrix one after the other, all kept alive (so that each search runs on other memory), the tuner's report
of each, and the SpMV inside a short CG run on the last two.  Informative on a device where the first upload ends above the fast
level; on the others it shows that later uploads, with 1-3 matrices already resident, still find it.  usage: placement_lab15.py [uploads=4]"""
import os
import sys

os.environ.setdefault("SB_PLACE_REPORT", "1")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

uploads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = capi.init(0)
probs = []
for k in range(uploads):
    p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
    assert p.use_packed(0) == 0
    probs.append(p)
    print("upload %d: %r" % (k, p.placement_report()), flush=True)
for k, p in enumerate(probs):
    cg = hostapi.CG(p)
    cg.start(itermax=1202, eps=0.0)
    cg.run_iters(600)  # (settle: DESIGN 4.1, the time effect)
    cg.spmv_timing(True)
    cg.run_iters(600)
    ms, n = cg.spmv_ms()
    cg.spmv_timing(False)
    cg.finish()
    cg.free()
    print("upload %d: SpMV inside CG %.1f us per launch" % (k, 1e3 * ms / max(n, 1)), flush=True)
