#!/usr/bin/env python3
"""Does the SpMV launch time of the CG loop move with TIME on fixed memory?  One process, one problem (the upload's tuner on, as in
the product), one CG object; after every sb_cg_start the loop runs in chunks of `chunk` bodies with HIP events around every SpMV
launch, and the chunk's mean launch time is printed: the series after a start shows whether the loop enters at one level and
settles at another (bench.py's K = 20 window against its sustained leg).  usage: placement_lab9.py [starts=4] [chunks=60] [chunk=10]"""
import os
import sys
import time

os.environ.setdefault("SB_PLACE_REPORT", "1")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

starts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
print("placement: %r" % (p.placement_report(),), flush=True)
cg = hostapi.CG(p)
for s in range(starts):
    cg.start(itermax=chunks * chunk + 2, eps=0.0)
    row, wall = [], []
    for c in range(chunks):
        cg.spmv_timing(True)
        t0 = time.perf_counter()
        cg.run_iters(chunk)
        L.sb_sync()
        wall.append(1e6 * (time.perf_counter() - t0) / chunk)
        ms, n = cg.spmv_ms()
        row.append(1e3 * ms / max(n, 1))
    cg.spmv_timing(False)
    cg.finish()
    print("start %d: SpMV us per chunk of %d: %s" % (s, chunk, " ".join("%.0f" % v for v in row)), flush=True)
    print("start %d: wall us per body      : %s" % (s, " ".join("%.0f" % v for v in wall)), flush=True)
    if s % 2 == 1:
        time.sleep(0.5)  # (an idle gap between starts, as between bench.py's passes)
cg.free()
