#!/usr/bin/env python3
"""The reference-layout CG loop has two speeds (~142 and ~155 us per step at 128^3).  Is the speed a property of the
PROCESS or of where the buffers of one upload landed?  Several uploads of the same matrix in one process, each timed the same way
(480 clean steps in segments of 120, best of 3), and -- with `hold` -- with the previous uploads kept alive so that every upload
lands somewhere else.  usage: two_speeds.py [uploads=6] [hold]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

L = capi.init(0)
n_up = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hold = len(sys.argv) > 2 and sys.argv[2] == "hold"
kept = []
for u in range(n_up):
    p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
    assert p.use_packed(0) == 0
    cg = hostapi.CG(p)
    best = 1e9
    for _ in range(3):
        tot = 0.0
        for seg in range(4):
            cg.start(itermax=128, eps=0.0)
            cg.run_iters(6)
            L.sb_sync()
            t0 = time.perf_counter()
            cg.run_iters(120)
            L.sb_sync()
            tot += time.perf_counter() - t0
            cg.finish()
        best = min(best, tot / 480)
    print("upload %d: %.2f us per step (%.0f it/s)" % (u, 1e6 * best, 1.0 / best), flush=True)
    cg.free()
    if hold:
        kept.append(p)
    else:
        p.free()
