#!/usr/bin/env python3
"""cg_penalty_lab.py [n] [sigma] -- why is the reference-layout SpMV slower inside CG than stand-alone?
Times spmv_scs64 (HIP events around each launch, with the fused dot, as CG launches it) when the launch is preceded by
  a) nothing (back-to-back launches, x untouched)
  b) a read-only pass over 2 vectors (dot partials: 33.5 MB read)
  c) waxpby rewriting x (as the p update does to p: 33.5 MB read, 16.8 MB written)
  d) two waxpby (x and another vector rewritten: 33.5 MB written -- the p update's p and x)
  e) four waxpby (67 MB written: everything a CG body dirties)
so that the time can be read against the bytes the predecessor leaves to be written to HBM."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = capi.init(0)
p = hostapi.Problem("generate", n, n, n, fmt="scs", Cc=64, sigma=sigma)
assert p.use_packed(mode) == mode
x, y = DeviceVector.from_host(np.ones(p.nc)), DeviceVector(p.nr)
v = [DeviceVector.from_host(np.ones(p.nc)) for _ in range(3)]
q = DeviceVector.from_host(np.zeros(4 * ((p.nr + 255) // 256) + 4))
reps = 60
evs = [(L.sb_event_create(), L.sb_event_create()) for _ in range(reps)]


def pre(kind):
    if kind == "b":
        L.sb_ddot_partials(p.nr, v[0].ptr, v[1].ptr, q.ptr)
    if kind in "cde":
        L.sb_waxpby(p.nr, 1.0, x.ptr, 0.0, x.ptr, x.ptr)
    if kind in "de":
        L.sb_waxpby(p.nr, 1.0, v[0].ptr, 0.0, v[0].ptr, v[0].ptr)
    if kind == "e":
        L.sb_waxpby(p.nr, 1.0, v[1].ptr, 0.0, v[1].ptr, v[1].ptr)
        L.sb_waxpby(p.nr, 1.0, v[2].ptr, 0.0, v[2].ptr, v[2].ptr)


labels = {"a": "back to back, x untouched", "b": "after a read-only pass (33.5 MB read)", "c": "after x rewritten (16.8 MB written)",
          "d": "after 2 vectors rewritten (33.5 MB written)", "e": "after 4 vectors rewritten (67 MB written)"}
print("kernel mode %d, %d^3 sigma %d, %s" % (mode, n, sigma, L.sb_version().decode()))
for kind in "abcdea":
    for _ in range(5):
        pre(kind)
        L.sb_spmv_native_dot(p.matrix, x.ptr, y.ptr, q.ptr)
    for a, b in evs:
        pre(kind)
        L.sb_event_record(a)
        L.sb_spmv_native_dot(p.matrix, x.ptr, y.ptr, q.ptr)
        L.sb_event_record(b)
    L.sb_sync()
    t = sorted(1e3 * L.sb_event_elapsed_ms(a, b) for a, b in evs)
    print("  %-48s median %.2f us  min %.2f  p90 %.2f" % (labels[kind], t[len(t) // 2], t[0], t[int(0.9 * len(t))]), flush=True)
