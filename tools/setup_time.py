#!/usr/bin/env python3
"""setup_time.py -- how long the upload of HPCG 128^3 takes (generator + Sell-C-sigma layout + all pack levels), with
and without level 6 (measured: 1.6 s / 0.8 s at sigma = 256, 0.9 s at sigma = 1)"""
import sys, time, os
sys.path.insert(0, os.getcwd())
from sparsebench_amd import capi, hostapi
L = capi.init(0)
for sig, env in ((256, None), (256, "5"), (1, None)):
    if env: os.environ["SB_PACK"] = env
    elif "SB_PACK" in os.environ: del os.environ["SB_PACK"]
    t = time.time()
    p = hostapi.Problem("generate", 128, 128, 128, fmt="scs", Cc=64, sigma=sig)
    print("sigma", sig, "SB_PACK", env, "Problem() %.2f s, setup_seconds %.2f, mode %d" % (time.time() - t, p.setup_seconds, p.pack_info()["mode"]), flush=True)
    p.free()
