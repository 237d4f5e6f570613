#!/usr/bin/env python3
"""The CG step of the section-8d loop moves in steps of ~7.7 us with the allocation pattern of a process (placement_lab2) while the
stand-alone SpMV does not: is it the distance between the loop's vectors?  They live in one allocation, SB_CG_VEC_PAD_KB apart
(sbhip_cg.inc.h: sb_cg_create); one process, one matrix, a sweep of that gap.  usage: placement_lab4.py [pads_kb ...]"""
import os
import sys
import time

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

pads = [int(v) for v in sys.argv[1:]] or [0, 4, 16, 64, 128, 256, 512, 1024, 2048, 49152, 65536, 0, 0, 0]
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
for pad in pads:
    os.environ["SB_CG_VEC_PAD_KB"] = str(pad)
    cg = hostapi.CG(p)
    best = 1e9
    for seg in range(2):
        cg.start(itermax=128, eps=0.0)
        cg.run_iters(6)
        L.sb_sync()
        t0 = time.perf_counter()
        cg.run_iters(120)
        L.sb_sync()
        best = min(best, (time.perf_counter() - t0) / 120)
        cg.finish()
    cg.start(itermax=128, eps=0.0)
    cg.run_iters(6)
    cg.phase_timing(True)
    cg.run_iters(120)
    L.sb_sync()
    ph = {k: round(v[0], 2) for k, v in cg.phase_us().items()}
    cg.phase_timing(False)
    cg.finish()
    import ctypes as C
    vp = (C.c_uint64 * 8)()
    L.sb_cg_debug_ptrs(cg.ptr, vp)
    print("gap between the loop's vectors %7d KB: %.2f us per CG step | with an event after every launch: %s | r at %x" % (pad, 1e6 * best, ph, vp[0]), flush=True)
    cg.free()
