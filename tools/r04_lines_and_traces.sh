O=gpurun_out/r04; mkdir -p $O
b() { local name=$1; shift; python3 bench.py "$@" > $O/r04_bench_$name.json 2>> $O/bench.err; echo "bench $name rc=$?"; }
b n1_128_scs_sigma256
b n1_as_the_driver_types_it --gpus 1 --steps 20 --warmup 5
b n1_128_scs_sigma1 --sigma 1 --no-cpu
b n1_64_scs_sigma1 --n 64 --sigma 1
b n1_256_scs_sigma256 --n 256 --steps 40 --warmup 5 --no-cpu
b n1_128_crs --fmt crs --no-cpu
b irregular --workload irregular --irr-sigmas 1,256 --steps 120
bash tools/r04_clean_traces.sh
