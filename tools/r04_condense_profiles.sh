#!/usr/bin/env bash
# condense what tools/r04_final_profiles.sh left under gpurun_out/ into profiles/ (run in the build container)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r04
H=$(cat $O/source_hash.txt)
for f in n1_128_scs_sigma256 n1_as_the_driver_types_it n1_128_scs_sigma1 n1_64_scs_sigma1 n1_256_scs_sigma256 n1_128_crs irregular; do cp $O/r04_bench_$f.json profiles/r04_bench_$f.json; done
cp $O/r04_clean_reference_kernel_stats.csv $O/r04_clean_structure_kernel_stats.csv profiles/
( echo "# the loop \`value\` is quoted on: SpMV streams the reference's arrays (bench.py --no-cpu --steps 240 --no-preflight --passes clean --loops reference)"; cat $O/r04_clean_reference_trace_summary.txt; echo; echo "# structure-exploiting loop (... --loops structure)"; cat $O/r04_clean_structure_trace_summary.txt ) > profiles/r04_clean_loop_trace_summary.txt
cp "$(find gpurun_out/prof/r04_hpcg128/kt -name '*kernel_stats.csv')" profiles/r04_hpcg128_kernel_stats.csv
python tools/summarize_prof2.py r04_hpcg128 > profiles/r04_hpcg128_pmc_summary.txt
cp "$(find gpurun_out/prof/r04_hpcg128_crs/kt -name '*kernel_stats.csv')" profiles/r04_hpcg128_crs_native_kernel_stats.csv
python tools/summarize_prof2.py r04_hpcg128_crs spmv > profiles/r04_hpcg128_crs_native_pmc_summary.txt
cp "$(find gpurun_out/prof/r04_irregular/kt -name '*kernel_stats.csv')" profiles/r04_irregular_kernel_stats.csv
python tools/summarize_prof2.py r04_irregular spmv > profiles/r04_irregular_pmc_summary.txt
cp $O/r04_irregular_read_sizes.txt profiles/ 2>/dev/null || true
python tools/make_pmc_traffic.py r04 "sbhip 0.5" hash=$H r04_hpcg128=hpcg_27pt_128^3_per_gpu_scs_C64_sigma256 r04_hpcg128_crs=hpcg_27pt_128^3_per_gpu_crs_C64_sigma256 r04_irregular=irregular_fe_80^3_nodes_crs > /dev/null
python - <<'PY'
import json
p='profiles/r04_pmc_traffic.json'
d=json.load(open(p))
d.get("irregular_fe_80^3_nodes_crs", {}).pop("spmv_scs64", None)  # (launches of both sigmas averaged together: not a per-workload figure)
json.dump(d,open(p,"w"),indent=1)
for w,v in d.items():
    if isinstance(v,dict):
        for k,e in v.items(): print(w,k,round(e['bytes_per_launch']/1e6,1),'MB', e['source_hash'])
PY
