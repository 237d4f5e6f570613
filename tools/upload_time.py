import time, os, sys
sys.path.insert(0, ".")
from sparsebench_amd import capi, hostapi
capi.init(0)
for place in ("1", "0", "1"):
    os.environ["SB_PLACE"] = place
    t0 = time.perf_counter()
    p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
    t1 = time.perf_counter()
    print("SB_PLACE=%s: Problem() took %.2f s; placement %s" % (place, t1 - t0, p.placement_report()), flush=True)
    p.free()
