"""pattern_lab.py -- CPU feasibility study for the joint (value, slot-delta) dictionary.

For a Sell-64-sigma matrix: emulate the tile windows of build_lds_windows (sbhip_matrix.inc.h),
compute every element's LDS slot, and count the distinct pairs
    (value bits, slot - slot of the row's first stored element)
over the whole matrix.  Few pairs (<= 255) mean one byte per element can replace
today's 1-byte value code + 2-byte slot.  Test infrastructure only (uses oracle/).
"""
import sys, os, collections
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import pyoracle as po

MERGE_GAP = 8


def study(n, sigma, Cc=64):
    g = po.GMatrix.generate(n, n, n)
    s = g.to_scs(Cc, sigma)
    cp, cl = s.chunkPtr.astype(np.int64), s.chunkLens.astype(np.int64)
    col, val = s.colInd.astype(np.int64), s.val
    o2n = s.oldToNewPerm.astype(np.int64)
    nr = s.nr
    bits = val.view(np.uint64)
    pad = (col == 0) & (bits == 0)
    colp = np.where(col < nr, o2n[np.minimum(col, nr - 1)], col)
    nT = (s.nChunks + 3) // 4
    pairs = collections.Counter()
    tile_sets = []
    maxwin = 0
    for t in range(nT):
        tcols, trow, tj, tbits, tpad = [], [], [], [], []
        for c in range(4 * t, min(4 * t + 4, s.nChunks)):
            L = cl[c]
            sl = slice(cp[c], cp[c] + L * 64)
            tcols.append(colp[sl]); tbits.append(bits[sl]); tpad.append(pad[sl])
            trow.append(np.tile(np.arange(64) + 64 * (c - 4 * t), L))
            tj.append(np.repeat(np.arange(L), 64))
        tcols, tbits, tpad = map(np.concatenate, (tcols, tbits, tpad))
        trow, tj = np.concatenate(trow), np.concatenate(tj)
        real = ~tpad
        u = np.unique(tcols[real])
        if u.size == 0:
            tile_sets.append(frozenset()); continue
        brk = np.nonzero(np.diff(u) > MERGE_GAP)[0]
        first = np.concatenate(([u[0]], u[brk + 1]))
        last = np.concatenate((u[brk], [u[-1]]))
        lens = last - first + 1
        lds = 1 + np.concatenate(([0], np.cumsum(lens)[:-1]))
        maxwin = max(maxwin, 1 + lens.sum())
        seg = np.searchsorted(first, tcols, side="right") - 1
        slot = np.where(real, lds[seg] + tcols - first[seg], 0)
        # row base: slot of the row's first stored element (j == 0)
        base = np.zeros(256, dtype=np.int64)
        j0 = tj == 0
        base[trow[j0]] = slot[j0]
        delta = np.where(real, slot - base[trow], -(1 << 30))
        key = np.stack([tbits.astype(np.int64), delta], 1)
        uk, cnt = np.unique(key, axis=0, return_counts=True)
        ks = [tuple(k) for k in uk.tolist()]
        for k, c in zip(ks, cnt.tolist()):
            pairs[k] += c
        tile_sets.append(frozenset(ks))
    top = set(k for k, _ in pairs.most_common(255))
    covered = sum(1 for ts in tile_sets if ts <= top)
    classes = []
    for ts in tile_sets:  # the greedy clustering of build_patterns (sbhip_matrix.inc.h)
        for i, c in enumerate(classes):
            if ts <= c:
                break
        else:
            for i, c in enumerate(classes):
                if len(c | ts) <= 255:
                    classes[i] = c | ts
                    break
            else:
                classes.append(set(ts))
    print(f"n={n} sigma={sigma}: tiles={nT} maxwin={maxwin} distinct pairs={len(pairs)} "
          f"max pairs per tile={max(len(t) for t in tile_sets)} classes={len(classes)} "
          f"tiles covered by top-255={covered} ({100.0 * covered / nT:.1f} %)")
    g.free()


if __name__ == "__main__":
    for n, sg in ((32, 1), (32, 256), (64, 1), (64, 256)) + (((128, 256),) if "--big" in sys.argv else ()):
        study(n, sg)
