#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/{kt,fetch,write}) into profiles/<tag>_*.
usage: tools/summarize_prof.py <prof_dir> <tag>"""
import collections
import csv
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out, exist_ok=True)
ks = os.path.join(src, "kt", "r1_kernel_stats.csv")
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(out, "%s_kernel_stats.csv" % tag))
lines = []
for name in ("fetch", "write"):
    path = os.path.join(src, name, "r1_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        lines.append("%-44s %-11s launches=%4d mean_KiB=%12.1f min=%12.1f max=%12.1f" % (
            k[-44:], c, len(v), sum(v) / len(v), min(v), max(v)))
if lines:
    with open(os.path.join(out, "%s_pmc_summary.txt" % tag), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per-launch values in KiB as\n"
                "# reported.  gfx950: FETCH_SIZE counts HALF the bytes of coalesced reads (guide: MI355X_MICROARCH\n"
                "# 'HBM'); calibration on kernels with known byte counts at 128^3 (nr = 2097152): dot_spans_k<3>\n"
                "# reads r and Ap = 2 x 16 MiB and reports ~16.4k KiB, cg_update_p reads r, p, x = 3 x 16 MiB and\n"
                "# reports ~24.6k KiB: factor 2.00.  WRITE_SIZE is exact.\n")
        f.write("\n".join(lines) + "\n")
path = os.path.join(src, "sq", "r1_counter_collection.csv")
if os.path.exists(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out, "%s_sq_counters.txt" % tag), "w") as f:
        f.write("# rocprofv3 --pmc (SQ block, own pass), per-launch means summed over the GPU; WAVE_CYCLES and\n"
                "# *_ACTIVE in units of 4 clocks; spmv_scs64_pat runs 32768 waves per launch at 128^3\n")
        for (k, c), v in sorted(agg.items()):
            if k.startswith("__amd") or "pack" in k or "pat_" in k or "remap" in k:
                continue
            f.write("%-44s %-22s launches=%4d mean=%.4g\n" % (k[-44:], c, len(v), sum(v) / len(v)))
print("wrote", sorted(os.listdir(out)))
