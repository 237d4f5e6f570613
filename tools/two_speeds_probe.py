#!/usr/bin/env python3
"""One fresh process of the section-8d loop (HPCG 128^3, Sell-64-256, reference-layout SpMV): a clean timing of 3 x 120
steps, printed as one line.  Run plain, or directly behind `rocprofv3 ... --` (tools/two_speeds_pmc.sh): the kernel trace then
gives this process's spmv_scs64 duration (fast kind ~117 us, slow kind ~128 us) next to the counters of the same launches.
usage: two_speeds_probe.py [tag] [segments=3]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "plain"
segs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
cg = hostapi.CG(p)
best, tot = 1e9, 0.0
for seg in range(segs):
    cg.start(itermax=128, eps=0.0)
    cg.run_iters(6)
    L.sb_sync()
    t0 = time.perf_counter()
    cg.run_iters(120)
    L.sb_sync()
    dt = time.perf_counter() - t0
    cg.finish()
    best = min(best, dt / 120)
    tot += dt
print("two_speeds_probe %s: best segment %.2f us per step, mean %.2f us per step" % (tag, 1e6 * best, 1e6 * tot / (120 * segs)), flush=True)
cg.free()
p.free()
