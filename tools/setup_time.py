import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsebench_amd import capi, hostapi
L = capi.init(0)
for fmt, sg in (("scs", 256), ("crs", 1)):
    t0 = time.perf_counter()
    p = hostapi.Problem("generate", 128, 128, 128, fmt=fmt, Cc=64, sigma=sg)
    L.sb_sync()
    t1 = time.perf_counter()
    print("%s 128^3: problem set-up %.2f s (host generate+partition+convert %.2f s)" % (fmt, t1 - t0, p.setup_seconds), flush=True)
    p.free()
