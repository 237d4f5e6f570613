#!/usr/bin/env python3
"""Is fast / slow a property of the stream's allocation, of the vectors' allocation, or of the PAIR?  ns stream allocations and na
arena allocations of one process (allocated alternately, all held), the tuner's proxy step for every pair.
usage: placement_lab7.py [ns=8] [na=16]"""
import os
import sys

import numpy as np

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
na = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = capi.init(0)
p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
ab = L.sb_placement_arena_bytes(p.matrix)
col_bytes = 240 << 20
S, A = [], []
for i in range(max(ns, na)):
    if i < ns:
        S.append(L.sb_malloc(col_bytes + (460 << 20)))
    for _ in range(2 if i < na // 2 else 0):
        A.append(L.sb_malloc(ab))
A = A[:na]
print("streams: " + " ".join("%x" % s for s in S))
print("arenas:  " + " ".join("%x" % a for a in A), flush=True)
grid = np.zeros((len(S), len(A)))
for i, s in enumerate(S):
    L.sb_matrix_place_at(p.matrix, s, s + col_bytes)
    for j, a in enumerate(A):
        grid[i, j] = L.sb_placement_probe(p.matrix, a)
    print("stream %d: " % i + " ".join("%6.1f" % v for v in grid[i]), flush=True)
L.sb_matrix_place_home(p.matrix)
fast = grid < 0.93 * grid.max()
print("fast pairs: %d of %d; per stream: %s; per arena: %s" % (int(fast.sum()), grid.size, fast.sum(axis=1).tolist(), fast.sum(axis=0).tolist()))
