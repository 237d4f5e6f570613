#!/usr/bin/env python3
"""Is the speed kind of the section-8d loop (tools/two_speeds_pmc.sh) a matter of WHERE the buffers of a process land?  One
process, several uploads, each behind dummy device allocations of different sizes (in front of the matrix, and between the matrix
and the CG vectors) so that matrix and vectors land at other physical / virtual offsets each time; everything is freed in
between.  Round 3 found all uploads of a process at that process's speed with identical allocation sequences; this varies the
sequence.  usage: two_speeds_inproc.py"""
import os
import sys
import time

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

L = capi.init(0)
MB = 1 << 20
for mb_a, mb_b in [(0, 0), (160, 0), (0, 160), (352, 0), (0, 352), (96, 96), (1024, 0), (0, 1024), (0, 0)]:
    d1 = L.sb_malloc(mb_a * MB) if mb_a else None
    p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
    assert p.use_packed(0) == 0
    d2 = L.sb_malloc(mb_b * MB) if mb_b else None
    cg = hostapi.CG(p)
    best = 1e9
    for seg in range(3):
        cg.start(itermax=128, eps=0.0)
        cg.run_iters(6)
        L.sb_sync()
        t0 = time.perf_counter()
        cg.run_iters(120)
        L.sb_sync()
        best = min(best, (time.perf_counter() - t0) / 120)
        cg.finish()
    print("two_speeds_inproc: %4d MB in front of the matrix, %4d MB between matrix and vectors: %.2f us per step" % (mb_a, mb_b, 1e6 * best), flush=True)
    cg.free()
    p.free()
    for d in (d1, d2):
        if d:
            L.sb_free(d)
