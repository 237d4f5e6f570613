import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import pyoracle as po
from sparsebench_amd import capi, hostapi
L = capi.init(0)
for n, sigma in ((16, 1), (32, 1), (32, 256), (128, 256)):
    prob = hostapi.Problem("generate", n, n, n, fmt="scs", Cc=64, sigma=sigma)
    print("n", n, "sigma", sigma, prob.pack_info(), flush=True)
    cg = hostapi.CG(prob)
    print("  launches per body", cg.launches_per_body(), flush=True)
    k = cg.solve(30, 0.0)
    rr, pap = cg.history()
    x = cg.solution()
    if n <= 32:
        o = po.cg(po.GMatrix.generate(n, n, n), itermax=30, fmt="scs", Cc=64, sigma=sigma, dot="tree", want_x=True)
        print("  k", k, o["k"], "rr equal", np.array_equal(rr, o["rr"]), "pAp equal", np.array_equal(pap, o["pAp"]), "x equal", np.array_equal(x, o["x"][0]), flush=True)
        if not np.array_equal(rr, o["rr"]):
            print(rr[:5], o["rr"][:5])
    else:
        print("  rr[:3]", rr[:3], flush=True)
    cg.free(); prob.free()
