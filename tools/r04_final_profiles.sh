#!/usr/bin/env bash
# Round-4 profile set (run on the GPU box from the repo root; results under gpurun_out/r04/, condensed into profiles/ by
# tools/r04_condense_profiles.sh in the build container):
#   PMC passes first (tools/prof_run.sh: one counter group per run) and the traffic file made from them on the box, then the bench
#   lines as fresh processes (N = 1: configs[2] with the default step count and as the driver types it, configs[1], sigma = 1,
#   CRS, 256^3, configs[4] stand-in), then rocprofv3 --kernel-trace --stats of the CLEAN loop of each kind
#   (tools/r04_clean_traces.sh: `--loops reference` = exactly what `value` is quoted on; `--loops structure`).
set -o pipefail
O=gpurun_out/r04
mkdir -p $O gpurun_out/prof
python3 -c "from sparsebench_amd import srchash; print(srchash.csrc_hash())" > $O/source_hash.txt
b() { local name=$1; shift; python3 bench.py "$@" > $O/r04_bench_$name.json 2>> $O/bench.err; echo "bench $name rc=$?"; }
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
# PMC passes of the default bench (both loops: spmv_scs64 and spmv_prog_fusep in one process)
tools/prof_run.sh r04_hpcg128 bench.py --no-cpu --steps 60 --warmup 5 --no-preflight --sustained-steps 0
# the native CRS kernel inside CG, and the irregular stand-in: kernel trace + FETCH / WRITE passes + the size split of the memory-side reads
for spec in "r04_hpcg128_crs:--fmt crs --loops reference --steps 40" "r04_irregular:--workload irregular --irr-sigmas 1,256 --steps 40"; do
  tag=${spec%%:*}; extra=${spec#*:}; out=gpurun_out/prof/$tag; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o r1 -- python3 bench.py --no-cpu --warmup 5 --no-preflight --passes clean,events --sustained-steps 0 $extra > $out/kt.log 2>&1 || echo "kt $tag failed"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o r1 -- python3 bench.py --no-cpu --warmup 5 --no-preflight --passes clean,events --sustained-steps 0 $extra > $out/fetch.log 2>&1 || echo "fetch $tag failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o r1 -- python3 bench.py --no-cpu --warmup 5 --no-preflight --passes clean,events --sustained-steps 0 $extra > $out/write.log 2>&1 || echo "write $tag failed"
done
# are the irregular CRS kernel's excess reads (PMC 1.11 x algorithmic) whole 128-B lines for 8-byte far gathers?
out=gpurun_out/prof/r04_irregular_sizes; mkdir -p $out
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/pmc -o r1 -- python3 tools/irregular_prof_run.py crs 1 80 20 > $out/pmc.log 2>&1 || echo "size-split pass failed"
python3 - <<'PY' | tee gpurun_out/r04/r04_irregular_read_sizes.txt
import csv, glob, collections
f = glob.glob("gpurun_out/prof/r04_irregular_sizes/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if "spmv_crs" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("irregular stand-in, spmv_crs_split, memory-side read requests per launch (median of %d launches):" % max([len(v) for v in agg.values()] + [0]))
for k, v in sorted(agg.items()):
    v = sorted(v); print("  %-28s %.4g" % (k, v[len(v) // 2]))
PY
# HBM bytes per launch from those passes -> profiles/r04_pmc_traffic.json ON THE BOX, so that the bench lines below (fresh processes)
# find a traffic figure measured with the very kernel sources they run
python3 tools/summarize_prof2.py r04_hpcg128 > $O/r04_hpcg128_pmc_summary.txt
python3 tools/summarize_prof2.py r04_hpcg128_crs spmv > $O/r04_hpcg128_crs_native_pmc_summary.txt
python3 tools/summarize_prof2.py r04_irregular spmv > $O/r04_irregular_pmc_summary.txt
python3 tools/make_pmc_traffic.py r04 "sbhip 0.5" hash=$(cat $O/source_hash.txt) r04_hpcg128=hpcg_27pt_128^3_per_gpu_scs_C64_sigma256 r04_hpcg128_crs=hpcg_27pt_128^3_per_gpu_crs_C64_sigma256 r04_irregular=irregular_fe_80^3_nodes_crs > /dev/null
python3 - <<'PY'
import json
p = "profiles/r04_pmc_traffic.json"
d = json.load(open(p))
d.get("irregular_fe_80^3_nodes_crs", {}).pop("spmv_scs64", None)  # (launches of both sigmas averaged together: not a per-workload figure)
json.dump(d, open(p, "w"), indent=1)
PY
cp profiles/r04_pmc_traffic.json $O/r04_pmc_traffic.json
b n1_128_scs_sigma256
b n1_as_the_driver_types_it --gpus 1 --steps 20 --warmup 5
b n1_128_scs_sigma1 --sigma 1 --no-cpu
b n1_64_scs_sigma1 --n 64 --sigma 1
b n1_256_scs_sigma256 --n 256 --steps 40 --warmup 5 --no-cpu
b n1_128_crs --fmt crs --no-cpu
b irregular --workload irregular --irr-sigmas 1,256 --steps 120
bash tools/r04_clean_traces.sh
echo done
