#!/usr/bin/env python3
"""The pair relation: ONE giant allocation, the stream at 16 positions in its lower half, the loop's vectors at 16 positions in its
upper half, the tuner's proxy step for all 256 pairs.  (The tuner's probe sequences show fast and slow pairs in long runs along
the allocation order, with the stream fixed as with the arena fixed: is there a rule?)  usage: placement_lab6.py [giant_gb=48] [step_gb=1.5]"""
import os
import sys

import numpy as np

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

giant_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 48.0
step_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
GB = 1 << 30
giant = L.sb_malloc(int(giant_gb * GB))
ab = L.sb_placement_arena_bytes(p.matrix)
col_bytes = 240 << 20  # (room for colInd: 224.1 MB + slack, 2 MiB aligned)
half = int(giant_gb * GB / 2) & ~((2 << 20) - 1)
n = 16
spos = [int(i * step_gb * GB) & ~((2 << 20) - 1) for i in range(n)]
apos = [half + (int(i * step_gb * GB) & ~((2 << 20) - 1)) for i in range(n)]
assert spos[-1] + (1 << 30) <= half and apos[-1] + ab <= int(giant_gb * GB)
grid = np.zeros((n, n))
print("rows: stream at giant + k x %.1f GB; columns: vectors at giant + %.1f GB + k x %.1f GB; giant = %x" % (step_gb, half / GB, step_gb, giant), flush=True)
for i, so in enumerate(spos):
    L.sb_matrix_place_at(p.matrix, giant + so, giant + so + col_bytes)
    for j, ao in enumerate(apos):
        grid[i, j] = L.sb_placement_probe(p.matrix, giant + ao)
    print("stream %5.1f GB: " % (so / GB) + " ".join("%6.1f" % v for v in grid[i]), flush=True)
L.sb_matrix_place_home(p.matrix)
print("min %.2f max %.2f; fast (< 0.93 x max) pairs: %d of %d" % (grid.min(), grid.max(), int((grid < 0.93 * grid.max()).sum()), grid.size))
