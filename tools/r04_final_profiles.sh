#!/usr/bin/env bash
# Round-4 profile set (run on the GPU box from the repo root; results under gpurun_out/r04/, condensed into profiles/ by
# tools/r04_condense_profiles.sh in the build container):
#   bench lines (N = 1: configs[2] with the default step count and as the driver types it, configs[1], sigma = 1, CRS, 256^3,
#   configs[4] stand-in), rocprofv3 --kernel-trace --stats of the CLEAN loop of each kind (`--loops reference` = exactly what
#   `value` is quoted on; `--loops structure`), and the PMC passes (tools/prof_run.sh: one counter group per run).
set -o pipefail
O=gpurun_out/r04
mkdir -p $O gpurun_out/prof
python3 -c "from sparsebench_amd import srchash; print(srchash.csrc_hash())" > $O/source_hash.txt
b() { local name=$1; shift; python3 bench.py "$@" > $O/r04_bench_$name.json 2>> $O/bench.err; echo "bench $name rc=$?"; }
b n1_128_scs_sigma256
b n1_as_the_driver_types_it --gpus 1 --steps 20 --warmup 5
b n1_128_scs_sigma1 --sigma 1 --no-cpu
b n1_64_scs_sigma1 --n 64 --sigma 1
b n1_256_scs_sigma256 --n 256 --steps 40 --warmup 5 --no-cpu
b n1_128_crs --fmt crs --no-cpu
b irregular --workload irregular --irr-sigmas 1,256 --steps 120
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
for spec in "reference:--loops reference" "structure:--loops structure"; do
  tag=${spec%%:*}; extra=${spec#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r04_clean_$tag -o r1 -- python3 bench.py --no-cpu --steps 240 --no-preflight --passes clean --sustained-steps 0 $extra > gpurun_out/prof/r04_clean_$tag.json 2> gpurun_out/prof/r04_clean_$tag.err || echo "clean trace $tag failed"
  f=$(find gpurun_out/prof/r04_clean_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_clean_${tag}_kernel_stats.csv
  t=$(find gpurun_out/prof/r04_clean_$tag -name "*kernel_trace.csv" | head -1); python3 tools/scalar_anatomy.py trace "$t" > $O/r04_clean_${tag}_trace_summary.txt
  cp gpurun_out/prof/r04_clean_$tag.json $O/r04_clean_${tag}_bench_line.json
  rm -rf gpurun_out/prof/r04_clean_$tag
done
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
echo done
