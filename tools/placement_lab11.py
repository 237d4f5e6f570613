#!/usr/bin/env python3
"""The CG loop's SpMV launch by launch on fixed memory: WHEN does it flip between its two levels (tools/placement_lab9.py), and
does the depth of the launch queue matter?  One process, one problem (tuner on), one CG object; per start either all bodies are
enqueued in one call (`deep`) or one body per call with a sync after each (`shallow`); every SpMV launch is bracketed by events and
the series is printed run-length encoded (level = below / above the series' midpoint).  usage: placement_lab11.py [iters=1500]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("SB_PLACE_REPORT", "1")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
print("placement: %r" % (p.placement_report(),), flush=True)
cg = hostapi.CG(p)
buf = (C.c_float * iters)()


def rle(us):
    mid = 0.5 * (np.percentile(us, 5) + np.percentile(us, 95))
    if np.percentile(us, 95) - np.percentile(us, 5) < 3.0:
        return "one level: %.1f us (5..95 %%: %.1f..%.1f)" % (np.median(us), np.percentile(us, 5), np.percentile(us, 95))
    lv = us > mid
    out, i = [], 0
    while i < len(us):
        j = i
        while j < len(us) and lv[j] == lv[i]:
            j += 1
        out.append("%s x%d (%.1f)" % ("SLOW" if lv[i] else "fast", j - i, us[i:j].mean()))
        i = j
    return " | ".join(out[:80]) + (" ..." if len(out) > 80 else "")


for name in ("deep", "shallow", "deep", "shallow", "deep", "deep"):
    cg.start(itermax=iters + 2, eps=0.0)
    cg.spmv_timing(True)
    if name == "deep":
        cg.run_iters(iters)
    else:
        for _ in range(iters):
            cg.run_iters(1)
            L.sb_sync()
    n = L.sb_cg_spmv_us_series(cg.ptr, buf, iters)
    cg.spmv_timing(False)
    cg.finish()
    us = np.array(buf[:min(n, iters)], dtype=np.float64)
    print("%-7s %d launches, mean %.1f us: %s" % (name, n, us.mean(), rle(us)), flush=True)
cg.free()
