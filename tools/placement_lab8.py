#!/usr/bin/env python3
"""Is there a TIME-dependent component on top of the placement?  One process, one stream, 12 spaced arena allocations; the tuner's
proxy step of every pair is taken again and again while the device is kept busy with the SpMV in between (~20 s of load), with the
memory temperature read from rocm-smi every few rounds.  If a pair's level drifts with time / temperature, the levels are not the
placement's alone.  usage: placement_lab8.py [rounds=40] [busy_launches=1500]"""
import os
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402
from sparsebench_amd.capi import DeviceVector  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
busy = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
ab = L.sb_placement_arena_bytes(p.matrix)
A, spacers = [], []
for i in range(12):
    A.append(L.sb_malloc(ab))
    spacers.append(L.sb_malloc(700 << 20))
dx, dy = DeviceVector(p.nc), DeviceVector(p.nr)


def temps():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showtemp", "--showclocks", "--showpower"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10).stdout.decode()
        keep = [ln.split(":", 1)[1].strip() for ln in out.splitlines() if ("Temperature" in ln or "mclk" in ln or "fclk" in ln or "Power" in ln) and ":" in ln]
        return " | ".join(keep)
    except Exception as e:
        return "rocm-smi: %s" % e


t0 = time.time()
print("t = 0: %s" % temps(), flush=True)
for r in range(rounds):
    v = [L.sb_placement_probe(p.matrix, a) for a in A]
    print("t = %5.1f s  " % (time.time() - t0) + " ".join("%6.1f" % x for x in v), flush=True)
    for _ in range(busy):
        L.sb_spmv_native(p.matrix, dx.ptr, dy.ptr)
    L.sb_sync()
    if r % 8 == 7:
        print("t = %5.1f s: %s" % (time.time() - t0, temps()), flush=True)
