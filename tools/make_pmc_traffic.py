#!/usr/bin/env python3
"""Collect the HBM bytes per launch of the SpMV kernels from the PMC passes (tools/prof_run.sh ->
tools/summarize_prof2.py -> gpurun_out/prof/<tag>/traffic.json) into profiles/<round>_pmc_traffic.json, the file
bench.py reads roofline.traffic from.  Every entry carries the content hash of the kernel sources it was collected
with (sparsebench_amd/srchash.py): bench.py uses an entry only when that matches the sources of the tree it runs from.
usage: make_pmc_traffic.py <round> "<library version>" [hash=<16 hex>] <tag>=<workload> [<tag>=<workload> ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebench_amd import srchash  # noqa: E402

rnd, version = sys.argv[1], sys.argv[2]
# the hash of the kernel sources the profiled library was built from: pass it (3rd argument "hash=<16 hex>", as printed by
# the profiled run on the GPU box) or let it default to the sources of THIS tree -- only right when nothing under
# sparsebench_amd/csrc/ changed since the profile was taken
src_hash = srchash.csrc_hash()
if len(sys.argv) > 3 and sys.argv[3].startswith("hash="):
    src_hash = sys.argv.pop(3)[5:]
out = {"_comment": "HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_run.sh); "
                   "bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 reports half the bytes of wide coalesced reads; calibration in "
                   "profiles/r01*_pmc_summary.txt).  bench.py copies bytes_per_launch into roofline.traffic when workload, kernel "
                   "and the kernel-source hash (sparsebench_amd/srchash.py) match.", "library_version": version, "source_hash": src_hash}
for spec in sys.argv[3:]:
    tag, workload = spec.split("=", 1)
    t = json.load(open(os.path.join(ROOT, "gpurun_out", "prof", tag, "traffic.json")))
    for k, e in t.items():
        name = k.split("<")[0].replace("sbk::", "").strip()
        if not name.startswith("spmv"):
            continue
        if name == "spmv_scs64_pat" and k.rstrip("> ").endswith("true") and k.count(",") == 4:
            name = "spmv_scs64_pat_masked"  # <CPT, DOT, SKIPPAD, HALO, MASKED = true>: the level-6 form (bench.py: mode 5)
        e = dict(e, library_version=version, source_hash=src_hash, source_tag=tag, kernel_instance=k)
        out.setdefault(workload, {})[name] = e
path = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % rnd)
json.dump(out, open(path, "w"), indent=1)
print(open(path).read())
