#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE ITSELF, compiled in place
(oracle/_ref, built by oracle/build_ref.sh from /root/reference/src).

Run in the build container only (the GPU box has no /root/reference); the JSON files
are committed.  What is captured:

  cg_hist_1rank.json   every r.r / p.Ap of solveCG, %.17e, strict-IEEE CRS build,
                       1 rank: band_klein, HPCG 8^3 16^3 32^3 64^3 (150 its),
                       128^3 (60 its)
  cg_hist_mpi.json     the same from the full MPI reference (mpiexec -n 2/4) on
                       HPCG 16^3 per rank, and band_klein on 2 ranks
  spmv_ref.json        spMVM(x = 1) of the CRS reference for test0..10 + band_klein
  scs_layout_fix.json  Sell-C-sigma arrays of the reference's convertMatrix with its
                       two C/sigma-clobbering assignments deleted (src/matrix-SCS.c:
                       42-43; see oracle/build_ref.sh) for test matrices at several
                       (C, sigma) incl. sigma > 1, plus its literal spMVM output

Data only: no reference source text is stored.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REFD = os.path.join(OUT, "ref")


def fl(a):
    return ["%.17e" % v for v in a]


def one_rank():
    ref = po.Ref("crs")
    out = {}
    ref.setup(os.path.join(REFD, "matrix_band_klein.mtx"))
    h = ref.solve_cg(150)
    out["band_klein"] = {"itermax": 150, "k": h["k"], "rr": fl(h["rr"]), "pAp": fl(h["pAp"])}
    for n, it in ((8, 150), (16, 150), (32, 150), (64, 150), (128, 60)):
        ref.setup("generate", n, n, n)
        h = ref.solve_cg(it)
        out["hpcg%d" % n] = {"itermax": it, "k": h["k"], "rr": fl(h["rr"]), "pAp": fl(h["pAp"])}
        print("hpcg", n, h["k"], len(h["rr"]), flush=True)
    json.dump(out, open(os.path.join(OUT, "cg_hist_1rank.json"), "w"), indent=0)


def mpi():
    exe = os.path.join(ROOT, "oracle", "_ref", "sb_ref_mpi")
    if not os.path.exists(exe):
        print("no MPI reference build; skipping cg_hist_mpi.json")
        return
    out = {}
    env = dict(os.environ, PATH="/opt/conda/bin:" + os.environ["PATH"])
    cases = [("hpcg16_x2", 2, ["-x", "16", "-y", "16", "-z", "16", "-i", "100"]),
             ("hpcg16_x4", 4, ["-x", "16", "-y", "16", "-z", "16", "-i", "100"]),
             ("hpcg8_x8", 8, ["-x", "8", "-y", "8", "-z", "8", "-i", "60"]),
             ("band_klein_x2", 2, ["-m", os.path.join(REFD, "matrix_band_klein.mtx"), "-i", "150"])]
    for name, nranks, args in cases:
        with tempfile.NamedTemporaryFile() as f:
            env["DDOT_LOG"] = f.name
            subprocess.check_call(["/opt/conda/bin/mpiexec", "-n", str(nranks), exe] + args, env=env,
                                  stdout=subprocess.DEVNULL)
            rr, pap = [], []
            for line in open(f.name):
                kind, v = line.split()
                (rr if kind == "rr" else pap).append(float(v))
        out[name] = {"ranks": nranks, "args": args[:-2] if "-m" not in args else ["-m", "band_klein"],
                     "itermax": int(args[-1]), "rr": fl(rr), "pAp": fl(pap)}
        print(name, len(rr), len(pap), flush=True)
    json.dump(out, open(os.path.join(OUT, "cg_hist_mpi.json"), "w"), indent=0)


def spmv():
    ref = po.Ref("crs")
    out = {}
    names = ["test%d" % i for i in range(11)] + ["matrix_band_klein"]
    for nm in names:
        ref.setup(os.path.join(REFD, nm + ".mtx"))
        y = ref.spmv(np.ones(ref.nc))
        out[nm] = fl(y)
    json.dump(out, open(os.path.join(OUT, "spmv_ref.json"), "w"), indent=0)


def scs_fix():
    ref = po.Ref("scs_fix")
    out = {}
    for nm in ["test%d" % i for i in range(11)]:
        for Cc, sg in ((1, 1), (2, 1), (4, 1), (2, 4), (4, 8), (8, 2), (3, 5)):
            ref.setup(os.path.join(REFD, nm + ".mtx"), Cc=Cc, sigma=sg)
            d = ref.scs()
            y = ref.spmv(np.arange(1, ref.nc + 1, dtype=np.float64), ny=d["nrPadded"])
            out["%s_C%d_s%d" % (nm, Cc, sg)] = {
                k: (v.tolist() if isinstance(v, np.ndarray) and v.dtype != np.float64 else
                    fl(v) if isinstance(v, np.ndarray) else int(v)) for k, v in d.items()}
            out["%s_C%d_s%d" % (nm, Cc, sg)]["y_literal_x_iota"] = fl(y)
    json.dump(out, open(os.path.join(OUT, "scs_layout_fix.json"), "w"), indent=0)


def bmx():
    """binary matrix files written by the reference's own `-c <file.mtx>` (src/main.c:41-52 ->
    src/matrixBinfile.c:38-104, built unmodified into oracle/_ref/sb_ref_mpi): tests/golden/ref/*.bmx"""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "sb_ref_mpi")
    if not os.path.exists(exe):
        print("bmx: no sb_ref_mpi (no MPI compiler wrapper) -- skipped")
        return
    with tempfile.TemporaryDirectory() as tmp:
        for nm in ("matrix_band_klein", "test0", "test8"):
            shutil.copy(os.path.join(REFD, nm + ".mtx"), tmp)
            subprocess.run([exe, "-c", os.path.join(tmp, nm + ".mtx")], cwd=tmp, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, check=False)  # exits through commAbort
            shutil.copy(os.path.join(tmp, nm + ".bmx"), os.path.join(OUT, "ref", nm + ".bmx"))


if __name__ == "__main__":
    po.build()
    one_rank()
    mpi()
    spmv()
    scs_fix()
    bmx()
