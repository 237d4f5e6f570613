#!/usr/bin/env bash
# condense what tools/r03_final_profiles.sh + tools/r03_rehearsals.sh left under gpurun_out/ into profiles/ (run in the build container)
set -e
cd "$(dirname "$0")/.."
H=$(cat gpurun_out/r03_source_hash.txt)
for f in r03_bench_n1_128_scs_sigma256 r03_bench_n1_as_the_driver_types_it r03_bench_n1_128_scs_sigma256_five_launches r03_bench_n1_128_scs_sigma256_separate_alpha_step r03_bench_n1_128_scs_sigma1 r03_bench_n1_64_scs_sigma1 r03_bench_n1_256_scs_sigma256 r03_bench_n1_128_crs r03_bench_irregular r03_bench_rehearsal_n2 r03_bench_rehearsal_n3 r03_bench_rehearsal_n4 r03_bench_rehearsal_n6 r03_bench_rehearsal_n2_128 r03_bench_rehearsal_n4_128; do cp gpurun_out/$f.json profiles/$f.json; done
cp gpurun_out/r03_clean_default_kernel_stats.csv gpurun_out/r03_clean_reflayout_kernel_stats.csv profiles/
( echo "# default loop (bench.py --no-cpu --steps 240 --no-preflight --passes clean)"; cat gpurun_out/r03_clean_default_trace_summary.txt; echo; echo "# reference-layout loop (... --pack-mode 0)"; cat gpurun_out/r03_clean_reflayout_trace_summary.txt ) > profiles/r03_clean_loop_trace_summary.txt
cp "$(find gpurun_out/prof/r03_hpcg128/kt -name '*kernel_stats.csv')" profiles/r03_hpcg128_kernel_stats.csv
python tools/summarize_prof2.py r03_hpcg128 > profiles/r03_hpcg128_pmc_summary.txt
cp "$(find gpurun_out/prof/r03_hpcg128_crs/kt -name '*kernel_stats.csv')" profiles/r03_hpcg128_crs_native_kernel_stats.csv
python tools/summarize_prof2.py r03_hpcg128_crs spmv > profiles/r03_hpcg128_crs_native_pmc_summary.txt
cp "$(find gpurun_out/prof/r03_irregular/kt -name '*kernel_stats.csv')" profiles/r03_irregular_kernel_stats.csv
python tools/summarize_prof2.py r03_irregular spmv > profiles/r03_irregular_pmc_summary.txt
python tools/make_pmc_traffic.py r03 "sbhip 0.4" hash=$H r03_hpcg128=hpcg_27pt_128^3_per_gpu_scs_C64_sigma256 r03_hpcg128_crs=hpcg_27pt_128^3_per_gpu_crs_C64_sigma256 r03_irregular=irregular_fe_80^3_nodes_crs > /dev/null
python - <<'PY'
import json
p='profiles/r03_pmc_traffic.json'
d=json.load(open(p))
d["irregular_fe_80^3_nodes_crs"].pop("spmv_scs64", None)  # (launches of both sigmas averaged together: not a per-workload figure)
json.dump(d,open(p,"w"),indent=1)
for w,v in d.items():
    if isinstance(v,dict):
        for k,e in v.items(): print(w,k,round(e['bytes_per_launch']/1e6,1),'MB', e['source_hash'])
PY
