"""Row slices of the one-rank irregular stand-in (host/sbh_irregular.c) as oracle matrices with GLOBAL column ids: what
every rank of a P-rank run starts from (file rule of src/comm.c:35-38).  Shared by the gloo and the GPU multi-rank workers."""
import numpy as np

from oracle import pyoracle as po
from sparsebench_amd import hostapi


def irregular_locs(n, size):
    one = hostapi.Problem("irregular", n, n, n, fmt="crs", rank=0, size=1, upload=False)
    rp = one.array("rowPtr").astype(np.int64)
    col, val = one.gm_entries()
    nr = one.nr
    one.free()
    locs = []
    base, extra = nr // size, nr % size
    for r in range(size):
        lo = r * base + min(r, extra)
        hi = lo + base + (1 if r < extra else 0)
        g = po.GMatrix.from_csr((rp[lo:hi + 1] - rp[lo]).astype(np.uint32), col[rp[lo]:rp[hi]], val[rp[lo]:rp[hi]], nc=nr)
        g.s.startRow, g.s.stopRow, g.s.totalNr = lo, hi - 1, nr
        locs.append(g)
    return locs
