#!/usr/bin/env python3
"""Is a device WITHOUT fast pairs slower as a device?  Per process: the raw streaming-read rate (sb_debug_stream_read_gbs on fresh
1 GiB / 4 GiB buffers, several allocations), the stand-alone SpMV on the uploaded matrix, then what the upload's tuner saw.
Run on several boxes and compare the ones with and without a fast level.  usage: placement_lab13.py"""
import os
import sys
import time

os.environ.setdefault("SB_PLACE_REPORT", "1")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

L = capi.init(0)
print("device: %s" % L.sb_device_name().decode(), flush=True)
for size, name in ((1 << 30, "1 GiB"), (4 << 30, "4 GiB")):
    vals = [L.sb_debug_stream_read_gbs(size, 20) for _ in range(4)]
    print("raw streaming read, %s buffer, 4 allocations: %s GB/s" % (name, " ".join("%.0f" % v for v in vals)), flush=True)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
print("placement: %r" % (p.placement_report(),), flush=True)
dx, dy = DeviceVector(p.nc), DeviceVector(p.nr)
for _ in range(50):
    L.sb_spmv_native(p.matrix, dx.ptr, dy.ptr)
L.sb_sync()
t0 = time.perf_counter()
for _ in range(400):
    L.sb_spmv_native(p.matrix, dx.ptr, dy.ptr)
L.sb_sync()
print("stand-alone SpMV (x, y in fresh allocations), 400 launches back to back: %.1f us per launch" % (1e6 * (time.perf_counter() - t0) / 400), flush=True)
for size, name in ((1 << 30, "1 GiB"),):
    vals = [L.sb_debug_stream_read_gbs(size, 20) for _ in range(4)]
    print("raw streaming read again, %s: %s GB/s" % (name, " ".join("%.0f" % v for v in vals)), flush=True)
