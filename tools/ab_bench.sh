#!/usr/bin/env bash
# A/B the SpMV variants inside CG on ONE device (same gpurun call): prints it/s and the
# in-situ SpMV launch time for each "ENV=..." argument.  Usage: tools/ab_bench.sh [bench args --] VAR=1 ...
cd "$(dirname "$0")/.."
extra=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do extra+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
for v in "$@"; do
  printf "%-34s" "$v"
  env $v python bench.py --no-cpu "${extra[@]}" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
q=d.get('roofline_reference_layout')
print('it/s=%.0f ms/step=%.4f spmv_us=%.1f GB/s=%.0f moved=%.0fGB/s %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['achieved'], r.get('moved_GBs',0), ('| ref-layout: %.1fus %.0fGB/s %.0fit/s' % (q['avg_launch_us'], q['achieved'], q['cg_iterations_per_s'])) if q else ''))"
done
