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
                "# 'HBM'); calibrated here on cg_update_p (2 x 16 MiB read -> 16397 KiB reported) and\n"
                "# cg_update_xr_dot (4 x 16 MiB -> 32786 KiB): factor 2.00 for 4-, 8- and 16-byte-per-lane loads.\n"
                "# WRITE_SIZE is exact.\n")
        f.write("\n".join(lines) + "\n")
print("wrote", os.listdir(out))
