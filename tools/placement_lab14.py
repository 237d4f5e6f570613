#!/usr/bin/env python3
"""When a process's loop runs at the slow state (SpMV 124-128 us) on a pair whose tuner proxy read the fast level: is the PROXY
slow too at that moment (then it is time's), or does the proxy still read ~131 us on the very same vectors (then the proxy does not
see what the loop sees)?  One process, tuner on; alternately: 300 loop bodies with events around every SpMV launch (mean launch
time), then the proxy step re-timed from the host on the CG object's own r / p / Ap (3 launches per body through the public ops, 40
bodies, wall clock around a sync), between solves.  usage: placement_lab14.py [rounds=12]"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("SB_PLACE_REPORT", "1")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
print("placement: %r" % (p.placement_report(),), flush=True)
cg = hostapi.CG(p)
ptrs = (C.c_uint64 * 8)()
L.sb_cg_debug_ptrs(cg.ptr, ptrs)
r, pp, ap, x = ptrs[0], ptrs[1], ptrs[3], ptrs[4]
n = p.nr


def proxy(with_x):
    def body():
        L.sb_waxpby(n, 1.0, r, 0.5, pp, pp)
        L.sb_spmv_native(p.matrix, pp, ap)
        if with_x:
            L.sb_waxpby(n, 1.0, x, 1e-3, pp, x)
        L.sb_waxpby(n, 1.0, r, -1e-3, ap, r)
    for _ in range(4):
        body()
    L.sb_sync()
    t0 = time.perf_counter()
    for _ in range(40):
        body()
    L.sb_sync()
    return 1e6 * (time.perf_counter() - t0) / 40


for k in range(rounds):
    cg.start(itermax=302, eps=0.0)
    cg.spmv_timing(True)
    t0 = time.perf_counter()
    cg.run_iters(300)
    L.sb_sync()
    wall = 1e6 * (time.perf_counter() - t0) / 300
    ms, cnt = cg.spmv_ms()
    cg.spmv_timing(False)
    cg.finish()
    a, b = proxy(False), proxy(True)
    print("round %2d: loop %.1f us per body, SpMV %.1f us per launch | proxy step on the same vectors %.1f us, with the x update %.1f us"
          % (k, wall, 1e3 * ms / max(cnt, 1), a, b), flush=True)
    if k % 4 == 3:
        time.sleep(0.4)
cg.free()
