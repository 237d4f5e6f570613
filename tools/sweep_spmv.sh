#!/usr/bin/env bash
# Sweep the SCS C=64 SpMV kernel variants through the C driver's -t spmv mode.
# Prints the reference-convention MB/s line (12*27*nr bytes per SpMV) per variant.
set -u
cd "$(dirname "$0")/../sparsebench_amd/bin"
N=${1:-128}
for nt in 1 0; do
  for u in 1 2 4 8 9; do
    r=$(SB_SCS_UNROLL=$u SB_SCS_NT=$nt ./sparseBench-SCS-HIP -x $N -y $N -z $N -i 300 -t spmv -C 64 -s ${2:-1} | grep "spMVM:")
    echo "n=$N sigma=${2:-1} unroll=$u nt=$nt $r"
  done
done
