#!/usr/bin/env bash
# usage: tools/prof_pipes.sh <tag> <python-script-and-args...>  -- which issue pipe is busy? (SQ busy / active counters)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof/$tag
mkdir -p $out
rocprofv3 -L > $out/counters_list.txt 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAVES \
  --output-format csv -d $out/pipes -o r1 -- python3 "$@" > $out/pipes.log 2>&1 || echo "pipes pass failed"
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $out/pipes2 -o r1 -- python3 "$@" > $out/pipes2.log 2>&1 || echo "pipes2 pass failed"
python3 - <<PY
import csv, collections, glob
for sub in ("pipes", "pipes2"):
    for path in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void sbk::", "")
            if "spmv" in k or "cg_update" in k or "dot_spans" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(agg):
            print(k)
            for c in sorted(agg[k]):
                v = agg[k][c]
                print("   %-26s launches=%4d mean=%14.1f" % (c, len(v), sum(v) / len(v)))
PY
