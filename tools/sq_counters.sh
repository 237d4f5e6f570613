#!/usr/bin/env bash
# usage (on the GPU box): tools/sq_counters.sh <out-tag> [env VAR=..] -- python3 tools/pat_lab.py ...
# SQ instruction counters per wave for spmv_scs64_pat (rocprofv3 --pmc, own pass, no tracing)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES \
  --output-format csv -d gpurun_out/prof/$tag -o r1 -- "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/prof/$tag/r1_counter_collection.csv")):
    if "spmv_scs64_pat" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$tag: " + "  ".join("%s/wave=%.0f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), sum(v) / len(v) / 32768) for k, v in sorted(agg.items())))
PY
