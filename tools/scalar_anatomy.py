#!/usr/bin/env python3
"""scalar_anatomy.py -- where do the ~4.5 us of a CG scalar step go?  

  part 1 (in-kernel, needs labs/libsbhip_scalar.so from tools/make_scalar_lab.py, loaded through SBHIP_LIBRARY):
          wall_clock64 stamps of thread 0: kernel entry -> partial loads back -> reduction done -> stores acknowledged
  part 2 (rocprofv3 --kernel-trace CSV of the same loop, path as argv[2]): per launch of every kernel of the loop its
          duration (begin -> end as the profiler sees it) and the gap to the previous kernel's end: dispatch + write-back

usage:  SBHIP_LIBRARY=$PWD/labs/libsbhip_scalar.so python tools/scalar_anatomy.py stamps [n] [sigma]
        python tools/scalar_anatomy.py trace <kernel_trace.csv>
"""
import csv
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stamps(n, sigma):
    from sparsebench_amd import capi, hostapi
    L = capi.init(0)
    p = hostapi.Problem("generate", n, n, n, fmt="scs", Cc=64, sigma=sigma)
    cg = hostapi.CG(p)
    cg.solve(150, 0.0)
    cg.solve(150, 0.0)
    out = np.zeros(4096 * 4, dtype=np.int64)
    L.sb_lab_sprof.restype = C.c_uint
    cnt = L.sb_lab_sprof(out.ctypes.data_as(C.c_void_p))
    q = out.reshape(4096, 4)[:min(cnt, 4096)]
    mode = (q[:, 3] >> 60) & 3
    t = q.copy()
    t[:, 3] &= (1 << 60) - 1
    print("%d scalar launches recorded (HPCG %d^3 sigma %d); wall_clock64 = 100 MHz, so 0.01 us resolution" % (cnt, n, sigma))
    for m, name in ((2, "alpha step (reads 4 x 8192 level-0 partials of p.Ap = 262 KB)"), (1, "beta step (reads 8192 level-1 values of r.r = 64 KB)")):
        s = t[mode == m].astype(np.float64) / 100.0
        if not len(s):
            continue
        d1, d2, d3 = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
        print("  %s: %d launches" % (name, len(s)))
        for lab, d in (("entry -> this thread's partial loads back", d1), ("-> butterfly, LDS, barrier, 16 wave sums", d2), ("-> control block stores acknowledged", d3),
                       ("entry -> stores acknowledged (in-kernel total)", s[:, 3] - s[:, 0])):
            print("      %-48s median %.2f us  p10 %.2f  p90 %.2f" % (lab, np.median(d), *np.percentile(d, [10, 90])))


def trace(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    stats = {}
    for i in range(1, len(rows)):
        name = rows[i][2].split("(")[0]
        name = name.replace("void ", "").replace("sbk::", "")
        gap = (rows[i][0] - rows[i - 1][1]) / 1e3
        dur = (rows[i][1] - rows[i][0]) / 1e3
        if gap > 50:  # host-side gaps between timed segments
            continue
        stats.setdefault(name, []).append((gap, dur))
    print("%-70s %7s %9s %9s %9s" % ("kernel (rocprofv3 begin / end timestamps)", "count", "dur us", "gap us", "dur+gap"))
    for name, v in sorted(stats.items(), key=lambda kv: -len(kv[1])):
        a = np.array(v)
        if len(a) < 20:
            continue
        print("%-70s %7d %9.2f %9.2f %9.2f" % (name[:70], len(a), np.median(a[:, 1]), np.median(a[:, 0]), np.median(a.sum(1))))


if __name__ == "__main__":
    if sys.argv[1] == "stamps":
        stamps(int(sys.argv[2]) if len(sys.argv) > 2 else 128, int(sys.argv[3]) if len(sys.argv) > 3 else 256)
    else:
        trace(sys.argv[2])
