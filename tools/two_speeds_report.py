#!/usr/bin/env python3
"""Condense tools/two_speeds_pmc.sh: per profiled process the mean spmv_scs64 duration (kernel trace) and the per-launch mean of
every counter of its group, sorted by duration -- the counter that separates the fast kind from the slow kind stands out."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]


def kname(full):
    """'void spmv_scs64<4, true, true>(args)' -> 'spmv_scs64'"""
    return full.split("(")[0].split("<")[0].split()[-1].split("::")[-1]


rows = []
for d in sorted(glob.glob(os.path.join(out, "p*_*"))):
    if not os.path.isdir(d):
        continue
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not kt:
        continue
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        name = kname(r["Kernel_Name"])
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    if cc:
        for r in csv.DictReader(open(cc[0])):
            name = kname(r["Kernel_Name"])
            cnt[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows.append((os.path.basename(d), dur, cnt))


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


print("== plain processes (no profiler)")
try:
    print(open(os.path.join(out, "plain.txt")).read().rstrip())
except OSError:
    pass
print("== profiled processes: spmv_scs64 median duration [us] (launches), then per-launch medians of the group's counters")
for name, dur, cnt in sorted(rows, key=lambda t: med(t[1].get("spmv_scs64", [0]))):
    k = "spmv_scs64"
    line = "%-14s %8.2f us (%d)" % (name, med(dur.get(k, [])), len(dur.get(k, [])))
    for c, v in sorted(cnt.get(k, {}).items()):
        line += "  %s=%.4g" % (c.replace("_sum", ""), med(v))
    g = cnt.get(k, {}).get("GRBM_GUI_ACTIVE")
    if g and dur.get(k):
        line += "  | eff. clock %.0f MHz (GUI_ACTIVE / 8 / duration)" % (med(g) / 8.0 / med(dur[k]))
    print(line)
    for other in ("cg_update_r_k", "cg_update_p"):
        if other in dur:
            print("%-14s   %s %.2f us" % ("", other, med(dur[other])))
