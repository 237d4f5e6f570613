#!/usr/bin/env python3
"""Condense the passes of tools/prof_run.sh (gpurun_out/prof/<tag>/{kt,fetch,write,sq,sq2,tcc}) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.txt and (with --traffic workload=... ) an entry set
for profiles/<round>_pmc_traffic.json.   usage: summarize_prof2.py <tag> [kernel-substring ...]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
want = sys.argv[2:] or ["spmv", "dot_spans", "cg_update", "cg_scalar", "waxpby", "halo", "gather"]
src = os.path.join(ROOT, "gpurun_out", "prof", tag)
out = os.path.join(ROOT, "profiles")


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


ks = find("kt", "kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(out, "%s_kernel_stats.csv" % tag))
counters = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq", "sq2", "tcc"):
    path = find(sub, "counter_collection.csv")
    if not path:
        continue
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void sbk::", "")
        if any(w in k for w in want):
            counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = ["# rocprofv3 --pmc, one counter group per pass (tools/prof_run.sh), per-launch MEANS over the launches of the run.",
         "# FETCH_SIZE / WRITE_SIZE in KiB as reported; gfx950 reports HALF the bytes of wide coalesced reads (guide",
         "# MI355X_MICROARCH 'HBM'): hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (factor calibrated in profiles/r01*_pmc_summary.txt).",
         "# SQ_* summed over the GPU; *_CYCLES / WAIT / ACTIVE in units of 4 clocks."]
traffic = {}
for k in sorted(counters):
    c = counters[k]
    lines.append("")
    lines.append(k)
    for name in sorted(c):
        v = c[name]
        lines.append("  %-24s launches=%5d mean=%14.1f min=%14.1f max=%14.1f" % (name, len(v), sum(v) / len(v), min(v), max(v)))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f, w = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        b = (2.0 * f + w) * 1024.0
        lines.append("  => HBM bytes per launch (2*FETCH + WRITE): %.1f MB" % (b / 1e6))
        traffic[k] = {"fetch_KiB_reported": f, "fetch_correction": 2.0, "write_KiB": w, "bytes_per_launch": b}
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        h, m = sum(c["TCC_HIT_sum"]) / len(c["TCC_HIT_sum"]), sum(c["TCC_MISS_sum"]) / len(c["TCC_MISS_sum"])
        lines.append("  => L2 hit rate %.3f" % (h / max(h + m, 1)))
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        wc = sum(c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"])
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if nm in c:
                lines.append("  => %s / WAVE_CYCLES = %.3f" % (nm, sum(c[nm]) / len(c[nm]) / max(wc, 1)))
open(os.path.join(out, "%s_pmc_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(src, "traffic.json"), "w"), indent=1)
print("\n".join(lines))
