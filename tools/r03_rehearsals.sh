#!/usr/bin/env bash
# Round-3 rehearsals of `bench.py --gpus N` with N ranks sharing ONE GPU (host-staged transport for the collectives; the
# peer-mapped paths over HIP IPC on the same device): the flow, the pre-flight checks, every key of the line -- not a benchmark.
set -o pipefail
for n in 2 3 4 6; do
  python3 bench.py --gpus $n --transport host --n 48 --steps 20 --warmup 5 --no-cpu > gpurun_out/r03_bench_rehearsal_n$n.json 2> gpurun_out/r03_bench_rehearsal_n$n.err; echo "rehearsal n=$n rc=$?"
done
for n in 2 4; do
  python3 bench.py --gpus $n --transport host --steps 20 --warmup 5 --no-cpu > gpurun_out/r03_bench_rehearsal_n${n}_128.json 2> gpurun_out/r03_bench_rehearsal_n${n}_128.err; echo "rehearsal n=$n 128^3 rc=$?"
done
