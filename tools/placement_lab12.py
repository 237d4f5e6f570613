#!/usr/bin/env python3
"""Does the KIND of device memory change the placement levels -- in particular on a device that has no fast pair?  One process
(upload with SB_PLACE=0), `na` spaced arenas of plain hipMalloc memory, and the stream copied into memory of several kinds:
plain hipMalloc, hipDeviceMallocUncached, hipDeviceMallocFinegrained, hipMallocManaged (prefetched to the device),
hipDeviceMallocContiguous; then the ARENA of each kind with the stream in plain memory.  The tuner's proxy step for every pair.
usage: placement_lab12.py [na=8] [slabs_per_kind=3]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("SB_PLACE", "0")  # (a lab of the placement itself: the upload's tuner stays out of it)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sparsebench_amd import capi, hostapi  # noqa: E402

na = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per_kind = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = capi.init(0)
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMallocManaged.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemPrefetchAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemAdvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
hip.hipDeviceSynchronize.argtypes = []


def alloc(kind, nbytes):
    q = C.c_void_p()
    if kind == "plain":
        return L.sb_malloc(nbytes)
    if kind == "managed":
        rc = hip.hipMallocManaged(C.byref(q), nbytes, 1)
        if rc == 0:
            hip.hipMemAdvise(q, nbytes, 3, 0)  # hipMemAdviseSetPreferredLocation: device 0
            hip.hipMemPrefetchAsync(q, nbytes, 0, None)
            hip.hipDeviceSynchronize()
    else:
        flags = {"uncached": 0x3, "finegrained": 0x1, "contiguous": 0x4}[kind]
        rc = hip.hipExtMallocWithFlags(C.byref(q), nbytes, flags)
    if rc != 0 or not q.value:
        print("  (%s: allocation refused, rc %d)" % (kind, rc), flush=True)
        return None
    return q.value


p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=256)
assert p.use_packed(0) == 0
ab = L.sb_placement_arena_bytes(p.matrix)
col_bytes = 240 << 20
slab_bytes = col_bytes + (460 << 20)
A = []
for i in range(na):
    A.append(L.sb_malloc(ab))
    L.sb_malloc(700 << 20)  # spacer (held)
row = [L.sb_placement_probe(p.matrix, a) for a in A]
print("stream where the upload put it      : " + " ".join("%6.1f" % v for v in row), flush=True)
best_plain = min(row)
last_managed = None
for kind in ("plain", "uncached", "finegrained", "contiguous", "managed"):
    for k in range(per_kind):
        s = alloc(kind, slab_bytes)
        if s is None:
            break
        L.sb_matrix_place_at(p.matrix, s, s + col_bytes)
        row = [L.sb_placement_probe(p.matrix, a) for a in A]
        print("stream in %-12s slab %d (%x): " % (kind, k, s) + " ".join("%6.1f" % v for v in row), flush=True)
    if kind == "managed" and s is not None:
        last_managed = s
L.sb_matrix_place_home(p.matrix)
special = {}
for kind in ("uncached", "finegrained", "contiguous", "managed"):
    row = []
    for k in range(per_kind):
        a = alloc(kind, ab)
        if a is None:
            break
        special.setdefault(kind, []).append(a)
        row.append(L.sb_placement_probe(p.matrix, a))
    print("arena in %-12s (stream home): " % kind + " ".join("%6.1f" % v for v in row), flush=True)
if special.get("managed"):
    L.sb_matrix_place_at(p.matrix, last_managed, last_managed + col_bytes)
    print("arena in managed, stream in managed: " + " ".join("%6.1f" % L.sb_placement_probe(p.matrix, a) for a in special["managed"]), flush=True)
    L.sb_matrix_place_home(p.matrix)
