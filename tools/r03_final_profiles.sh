#!/usr/bin/env bash
# Round-3 profile set (run on the GPU box from the repo root; results under gpurun_out/, condensed into profiles/ afterwards):
#   bench lines (N = 1: configs[2] as the driver types it and with the default step count, configs[1], CRS, configs[4] stand-in),
#   rocprofv3 --kernel-trace --stats of the CLEAN loop only (`--passes clean`: exactly what `value` is quoted on) for the default
#   and the reference-layout kernel, and the PMC passes (tools/prof_run.sh: one counter group per run).
set -o pipefail
mkdir -p gpurun_out/prof
python3 -c "from sparsebench_amd import srchash; print(srchash.csrc_hash())" > gpurun_out/r03_source_hash.txt
python3 bench.py > gpurun_out/r03_bench_n1_128_scs_sigma256.json 2> gpurun_out/r03_bench_n1.err; echo "bench default rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_n1_as_the_driver_types_it.json 2>> gpurun_out/r03_bench_n1.err; echo "bench driver-like rc=$?"
python3 bench.py --fuse-p 0 --fuse-alpha 0 --fuse-beta 0 --no-cpu > gpurun_out/r03_bench_n1_128_scs_sigma256_five_launches.json 2>> gpurun_out/r03_bench_n1.err; echo "bench five launches rc=$?"
python3 bench.py --fuse-alpha 0 --no-cpu > gpurun_out/r03_bench_n1_128_scs_sigma256_separate_alpha_step.json 2>> gpurun_out/r03_bench_n1.err; echo "bench fuse-alpha 0 rc=$?"
python3 bench.py --sigma 1 --no-cpu > gpurun_out/r03_bench_n1_128_scs_sigma1.json 2>> gpurun_out/r03_bench_n1.err; echo "bench sigma 1 rc=$?"
python3 bench.py --n 64 --sigma 1 > gpurun_out/r03_bench_n1_64_scs_sigma1.json 2>> gpurun_out/r03_bench_n1.err; echo "bench 64 rc=$?"
python3 bench.py --n 256 --steps 40 --warmup 5 --no-cpu > gpurun_out/r03_bench_n1_256_scs_sigma256.json 2>> gpurun_out/r03_bench_n1.err; echo "bench 256 rc=$?"
python3 bench.py --fmt crs --no-cpu > gpurun_out/r03_bench_n1_128_crs.json 2>> gpurun_out/r03_bench_n1.err; echo "bench crs rc=$?"
python3 bench.py --workload irregular --irr-sigmas 1,256 --steps 120 > gpurun_out/r03_bench_irregular.json 2>> gpurun_out/r03_bench_n1.err; echo "bench irregular rc=$?"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
for spec in "default:" "reflayout:--pack-mode 0"; do
  tag=${spec%%:*}; extra=${spec#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r03_clean_$tag -o r1 -- python3 bench.py --no-cpu --steps 240 --no-preflight --passes clean --sustained-steps 0 $extra > gpurun_out/prof/r03_clean_$tag.json 2> gpurun_out/prof/r03_clean_$tag.err || echo "clean trace $tag failed"
  f=$(find gpurun_out/prof/r03_clean_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03_clean_${tag}_kernel_stats.csv
  t=$(find gpurun_out/prof/r03_clean_$tag -name "*kernel_trace.csv" | head -1); python3 tools/scalar_anatomy.py trace "$t" > gpurun_out/r03_clean_${tag}_trace_summary.txt
  rm -rf gpurun_out/prof/r03_clean_$tag
done
tools/prof_run.sh r03_hpcg128 bench.py --no-cpu --steps 60 --warmup 5 --no-preflight --sustained-steps 0
# the native CRS kernel (equal nonzero windows) inside CG, and the irregular stand-in: FETCH / WRITE passes only
for spec in "r03_hpcg128_crs:--fmt crs --pack-mode 0 --steps 40" "r03_irregular:--workload irregular --irr-sigmas 1,256 --steps 40"; do
  tag=${spec%%:*}; extra=${spec#*:}; out=gpurun_out/prof/$tag; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o r1 -- python3 bench.py --no-cpu --warmup 5 --no-preflight --passes clean,events --sustained-steps 0 $extra > $out/kt.log 2>&1 || echo "kt $tag failed"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o r1 -- python3 bench.py --no-cpu --warmup 5 --no-preflight --passes clean,events --sustained-steps 0 $extra > $out/fetch.log 2>&1 || echo "fetch $tag failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o r1 -- python3 bench.py --no-cpu --warmup 5 --no-preflight --passes clean,events --sustained-steps 0 $extra > $out/write.log 2>&1 || echo "write $tag failed"
done
echo done
