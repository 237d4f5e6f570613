#!/usr/bin/env python3
"""What moves when the CG loop's SpMV flips between ~118 and ~125 us ON FIXED MEMORY (tools/placement_lab9.py)?  The loop runs in
chunks with HIP events around every SpMV launch while a thread samples the device's gpu_metrics blob from sysfs (the SMU's table:
clocks, power, throttle residencies) every ~2 ms; afterwards every 16-bit / 32-bit field of the blob that VARIES is printed per
chunk next to the chunk's SpMV time, so that a clock or a throttle counter that follows the flips shows up without knowing the
table's layout on this kernel (the first sample is kept raw).  usage: placement_lab10.py [starts=3] [chunks=60] [chunk=10] [out=gpurun_out/lab10]"""
import ctypes
import glob
import os
import struct
import sys
import threading
import time

os.environ.setdefault("SB_PLACE_REPORT", "1")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

starts = int(sys.argv[1]) if len(sys.argv) > 1 else 3
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 60
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
out = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/lab10"

L = capi.init(0)
hip = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(64)
hip.hipDeviceGetPCIBusId(buf, 64, 0)
bus = buf.value.decode().lower()
cands = [("/sys/bus/pci/devices/%s/gpu_metrics" % bus)] + sorted(glob.glob("/sys/class/drm/card*/device/gpu_metrics"))
path = next((c for c in cands if os.path.exists(c) and os.access(c, os.R_OK)), None)
print("device %s, metrics file %s" % (bus, path), flush=True)
if path is None:
    sys.exit("no readable gpu_metrics")

samples, stop = [], False


def sampler():
    fd = os.open(path, os.O_RDONLY)
    while not stop:
        try:
            b = os.pread(fd, 4096, 0)
        except OSError:
            os.close(fd)
            fd = os.open(path, os.O_RDONLY)
            continue
        samples.append((time.perf_counter(), b))
        time.sleep(0.0015)
    os.close(fd)


p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
print("placement: %r" % (p.placement_report(),), flush=True)
cg = hostapi.CG(p)
th = threading.Thread(target=sampler, daemon=True)
th.start()
rows = []  # (start, chunk, t0, t1, spmv_us)
for s in range(starts):
    cg.start(itermax=chunks * chunk + 2, eps=0.0)
    for c in range(chunks):
        cg.spmv_timing(True)
        t0 = time.perf_counter()
        cg.run_iters(chunk)
        L.sb_sync()
        t1 = time.perf_counter()
        ms, n = cg.spmv_ms()
        rows.append((s, c, t0, t1, 1e3 * ms / max(n, 1)))
    cg.spmv_timing(False)
    cg.finish()
    time.sleep(0.3)
stop = True
th.join()
cg.free()

first = samples[0][1]
size, fmt, rev = struct.unpack_from("<HBB", first, 0)
print("gpu_metrics: %d bytes read, header size %d format %d content %d, %d samples" % (len(first), size, fmt, rev, len(samples)), flush=True)
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
with open(out + "_first_sample.hex", "w") as f:
    f.write(first.hex() + "\n")
n = min(len(b) for _, b in samples)
# 16-bit fields that vary (little endian, even offsets), without the obvious counters (those that only ever grow)
vary16 = []
for off in range(4, n - 1, 2):
    vals = [struct.unpack_from("<H", b, off)[0] for _, b in samples]
    if len(set(vals)) > 1 and 0xFFFF not in vals:
        vary16.append((off, vals))
print("varying 16-bit offsets: %s" % " ".join(str(o) for o, _ in vary16), flush=True)


def in_chunk(vals, t0, t1):
    sel = [v for (t, _), v in zip(samples, vals) if t0 <= t <= t1]
    return sel


with open(out + "_per_chunk.txt", "w") as f:
    f.write("# start chunk spmv_us | per varying 16-bit offset: mean over the samples inside the chunk\n")
    f.write("# offsets: %s\n" % " ".join(str(o) for o, _ in vary16))
    for (s, c, t0, t1, us) in rows:
        cells = []
        for off, vals in vary16:
            sel = in_chunk(vals, t0, t1)
            cells.append("%.0f" % (sum(sel) / len(sel)) if sel else "-")
        f.write("%d %2d %6.1f | %s\n" % (s, c, us, " ".join(cells)))
# which offsets follow the flips: correlation of the per-chunk mean with the chunk's SpMV time
import numpy as np  # noqa: E402

us = np.array([r[4] for r in rows])
print("chunks: %d, SpMV us min %.1f median %.1f max %.1f" % (len(rows), us.min(), np.median(us), us.max()), flush=True)
report = []
for off, vals in vary16:
    m = []
    for (s, c, t0, t1, _) in rows:
        sel = in_chunk(vals, t0, t1)
        m.append(sum(sel) / len(sel) if sel else np.nan)
    m = np.array(m)
    ok = ~np.isnan(m)
    if ok.sum() > 10 and m[ok].std() > 0 and us[ok].std() > 0:
        r = float(np.corrcoef(m[ok], us[ok])[0, 1])
        fast, slow = m[ok & (us < np.median(us) - 1.5)], m[ok & (us > np.median(us) + 1.5)]
        report.append((abs(r), off, r, m[ok].min(), m[ok].max(), fast.mean() if fast.size else np.nan, slow.mean() if slow.size else np.nan))
report.sort(reverse=True)
print("offset  corr(field, SpMV us)  min  max  mean over fast chunks  mean over slow chunks")
for _, off, r, lo, hi, fa, sl in report[:25]:
    print("%5d  %+.2f  %8.1f %8.1f  %10.2f %10.2f" % (off, r, lo, hi, fa, sl))
print("series: " + " ".join("%.0f" % v for v in us), flush=True)
