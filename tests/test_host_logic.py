"""Host-side logic of the product (generator, Matrix Market reader, CRS / Sell-C-sigma
layout, row split) against the oracle and the reference's fixtures.  CPU only: the
layouts are built without uploading (upload=False)."""
import os
import re

import numpy as np
import pytest

from conftest import REFDATA, load_json
from oracle import pyoracle as po
from sparsebench_amd import hostapi


@pytest.mark.parametrize("dims", [(8, 8, 8), (5, 7, 3), (1, 1, 1), (2, 1, 9), (16, 16, 16)])
@pytest.mark.parametrize("rank,size", [(0, 1), (1, 3), (2, 3)])
def test_generator_matches_oracle(dims, rank, size):
    for name in ("generate", "generate7P"):
        p = hostapi.Problem(name, *dims, fmt="crs", rank=0, size=1, upload=False) if size == 1 else None
        g = po.GMatrix.generate(*dims, rank=rank, size=size, use7pt=name == "generate7P")
        if p is None:
            # multi-rank bricks keep GLOBAL columns until commPartition; compare via a 1-rank
            # problem of the stacked grid restricted to this rank's rows
            big = hostapi.Problem(name, dims[0], dims[1], dims[2] * size, fmt="crs", upload=False)
            rp = big.array("rowPtr")
            lo, hi = g.startRow, g.stopRow + 1
            col, val = big.gm_entries()
            assert np.array_equal(rp[lo:hi + 1] - rp[lo], g.rowPtr)
            assert np.array_equal(col[rp[lo]:rp[hi]], g.col) and np.array_equal(val[rp[lo]:rp[hi]], g.val)
            big.free()
            continue
        col, val = p.gm_entries()
        assert np.array_equal(p.array("rowPtr"), g.rowPtr)
        assert np.array_equal(col, g.col) and np.array_equal(val, g.val)
        assert (p.nnz, p.totalNnz, p.totalNr, p.nnzTrue) == (g.nnz, g.totalNnz, g.totalNr, g.nnzTrue)
        b, xe = p.rhs()
        assert np.array_equal(b, g.rhs()) and np.all(xe == 1.0)
        p.free()


MTX = ["test%d" % i for i in range(11)] + ["matrix_band_klein"]


@pytest.mark.parametrize("name", MTX)
def test_mm_reader_and_layouts_match_oracle(name):
    path = os.path.join(REFDATA, name + ".mtx")
    g = po.GMatrix.from_mtx(path)
    p = hostapi.Problem(path, fmt="crs", upload=False)
    col, val = p.gm_entries()
    assert np.array_equal(p.array("rowPtr"), g.rowPtr) and np.array_equal(col, g.col)
    assert np.array_equal(val, g.val) and np.array_equal(p.array("crs_colInd"), g.col)
    assert np.array_equal(p.values(), g.val)
    b, xe = p.rhs()
    assert np.all(b == 1.0) and xe is None
    p.free()
    for Cc, sg in ((1, 1), (2, 1), (4, 1), (2, 4), (64, 1), (64, 7), (3, 5)):
        s = g.to_scs(Cc, sg)
        q = hostapi.Problem(path, fmt="scs", Cc=Cc, sigma=sg, upload=False)
        for f in ("nChunks", "nrPadded", "nElems", "C", "sigma"):
            assert getattr(q, f) == getattr(s, f), f
        for a, b_ in (("chunkPtr", "chunkPtr"), ("chunkLens", "chunkLens"), ("scs_colInd", "colInd"),
                      ("oldToNewPerm", "oldToNewPerm"), ("newToOldPerm", "newToOldPerm")):
            assert np.array_equal(q.array(a), getattr(s, b_)), (a, Cc, sg)
        assert np.array_equal(q.values(), s.val)
        q.free()


def test_symmetric_and_pattern_files(tmp_path):
    f = tmp_path / "sym.mtx"
    f.write_text("%%MatrixMarket matrix coordinate real symmetric\n% c\n4 4 5\n1 1 4.0\n2 1 -1.5\n"
                 "3 3 2.0\n4 2 7.25\n4 4 1.0\n")
    g = po.GMatrix.from_mtx(str(f))
    p = hostapi.Problem(str(f), fmt="crs", upload=False)
    col, val = p.gm_entries()
    assert p.nnzTrue == 7 and np.array_equal(col, g.col) and np.array_equal(val, g.val)
    assert np.array_equal(col, [0, 1, 0, 3, 2, 1, 3])
    p.free()
    f2 = tmp_path / "pat.mtx"
    f2.write_text("%%MatrixMarket matrix coordinate pattern general\n3 3 4\n3 1\n1 1\n2 2\n1 3\n")
    p = hostapi.Problem(str(f2), fmt="crs", upload=False)
    col, val = p.gm_entries()
    assert np.array_equal(col, [0, 2, 1, 0]) and np.all(val == 1.0)
    p.free()


def test_scs_layout_reference_fixture_via_product():
    """the reference's own golden layout files, through the PRODUCT's convert code"""
    for name in ("test0", "test8"):
        for Cc in (1, 2, 4):
            exp = {}
            for line in open(os.path.join(REFDATA, "%s_C_%d_sigma_1.in" % (name, Cc))):
                m = re.match(r"(\w+): (.*)", line)
                if m:
                    exp[m.group(1)] = np.array([float(v) for v in m.group(2).replace(" ", "").split(",") if v])
            q = hostapi.Problem(os.path.join(REFDATA, name + ".mtx"), fmt="scs", Cc=Cc, sigma=1, upload=False)
            for a, b_ in (("chunkPtr", "chunkPtr"), ("chunkLens", "chunkLens"), ("scs_colInd", "colInd"),
                          ("oldToNewPerm", "oldToNewPerm"), ("newToOldPerm", "newToOldPerm")):
                assert np.array_equal(q.array(a).astype(np.float64), exp[b_]), (name, Cc, a)
            assert np.array_equal(q.values(), exp["val"])
            q.free()


def test_hpcg_sizes_of_the_scope_table():
    """nnz = (3n-2)^3, SCS C=64 sigma=1 nElems = 3n(3n-2)^2 at n = 64"""
    q = hostapi.Problem("generate", 64, 64, 64, fmt="scs", Cc=64, sigma=1, upload=False)
    assert q.nr == 262144 and q.nnzTrue == 6859000 and q.nChunks == 4096 and q.nElems == 6931200
    q.free()


@pytest.mark.parametrize("name", ["matrix_band_klein", "test0", "test8"])
def test_bmx_binary_files_match_the_reference_writer_and_reload(name, tmp_path, monkeypatch):
    """.bmx (src/matrixBinfile.h:15-19): the product's writer is byte-identical to the reference's
    (golden files written by the reference's own `-c`, tests/golden/make_golden.py); the reader
    slices rows per rank like src/matrixBinfile.c:155-166; SB_BMX_FP64=1 round-trips fp64 bits"""
    import shutil
    mtx = tmp_path / (name + ".mtx")
    shutil.copy(os.path.join(REFDATA, name + ".mtx"), mtx)
    out = hostapi.convert_mtx_to_bmx(mtx)
    golden = open(os.path.join(REFDATA, name + ".bmx"), "rb").read()
    assert open(out, "rb").read() == golden
    assert golden[:22] == b"# SparseBench DataFile" and golden[22:24] == b"\0\0"

    ref = hostapi.Problem(str(mtx), fmt="crs", upload=False)
    rp, (col, val) = np.array(ref.array("rowPtr")), ref.gm_entries()
    # f32 file: columns exact, values rounded to float32 (the reference's FEntry)
    p = hostapi.Problem(out, fmt="crs", upload=False)
    c2, v2 = p.gm_entries()
    assert np.array_equal(p.array("rowPtr"), rp) and np.array_equal(c2, col)
    assert np.array_equal(v2, val.astype(np.float32).astype(np.float64))
    assert np.all(p.rhs()[0] == 1.0)
    p.free()
    # every rank reads its own row slice; together they are the whole matrix
    for size in (2, 3):
        rows, cols, vals = 0, [], []
        for rank in range(size):
            start, rp2, c2, v2 = hostapi.read_bmx_slice(out, rank, size)
            n = len(rp2) - 1
            assert start == rows and n == ref.nr // size + (1 if ref.nr % size > rank else 0)
            assert np.array_equal(rp2, rp[start:start + n + 1] - rp[start])
            rows += n
            cols.append(c2), vals.append(v2)
        assert rows == ref.nr and np.array_equal(np.concatenate(cols), col)
        assert np.array_equal(np.concatenate(vals), val.astype(np.float32).astype(np.float64))
    # fp64 extension: bit-exact reload, header byte 23 = '8'
    monkeypatch.setenv("SB_BMX_FP64", "1")
    out64 = hostapi.convert_mtx_to_bmx(mtx)
    raw = open(out64, "rb").read()
    assert raw[23:24] == b"8" and len(raw) == 24 + 8 + 4 * (ref.nr + 1) + 16 * len(val)
    p = hostapi.Problem(out64, fmt="crs", upload=False)
    c3, v3 = p.gm_entries()
    assert np.array_equal(c3, col) and np.array_equal(v3.view(np.uint64), val.view(np.uint64))
    p.free(), ref.free()
