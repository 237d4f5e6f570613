#!/usr/bin/env bash
# The C host side (partitioner, halo-plan builder, converters, readers, .bmx files, irregular generator) under
# AddressSanitizer + UndefinedBehaviorSanitizer on the CPU: builds an instrumented libsparsebench_host.so into /tmp, runs
# tests/test_host_logic.py and tests/test_dist_gloo.py (1-8 processes) against it, restores the product library.
# (GPU sanitizers are not available on the pool; the HIP layer is covered by the bit-exact tests instead.)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/sb_asan
( cd sparsebench_amd/host && gcc -O1 -g -std=gnu11 -fPIC -Wall -Wextra -Wno-unused-parameter -fopenmp -fsanitize=address,undefined \
    -fno-omit-frame-pointer -I../../include -shared -o /tmp/sb_asan/libsparsebench_host.so sbh_base.c sbh_setup.c sbh_comm.c \
    sbh_convert.c sbh_solver.c sbh_flat.c sbh_binfile.c sbh_irregular.c -L../lib -lsbhip -Wl,-rpath,"$PWD/../lib" -lm )
cp sparsebench_amd/lib/libsparsebench_host.so /tmp/sb_asan/host_orig.so
trap 'cp /tmp/sb_asan/host_orig.so sparsebench_amd/lib/libsparsebench_host.so' EXIT
cp /tmp/sb_asan/libsparsebench_host.so sparsebench_amd/lib/libsparsebench_host.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_host_logic.py tests/test_dist_gloo.py -x -q
