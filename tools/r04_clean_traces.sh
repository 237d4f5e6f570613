#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of the CLEAN loop of each kind (`--loops reference` = exactly what `value` is quoted on;
# `--loops structure`), run on the GPU box from the repo root; results under gpurun_out/r04/.  The sustained leg (4800 steps in one
# go) stays in: it comes first and absorbs the device's first 50-150 ms under load (profiles/r04_placement_lab9.txt), as in a
# plain bench.py run; its launches are the same kernels of the same loop and are part of the averages.
set -o pipefail
O=gpurun_out/r04
mkdir -p $O gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
for spec in "reference:--loops reference" "structure:--loops structure"; do
  tag=${spec%%:*}; extra=${spec#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r04_clean_$tag -o r1 -- python3 bench.py --no-cpu --steps 240 --no-preflight --passes clean $extra > gpurun_out/prof/r04_clean_$tag.json 2> gpurun_out/prof/r04_clean_$tag.err || echo "clean trace $tag failed"
  f=$(find gpurun_out/prof/r04_clean_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_clean_${tag}_kernel_stats.csv
  t=$(find gpurun_out/prof/r04_clean_$tag -name "*kernel_trace.csv" | head -1); python3 tools/scalar_anatomy.py trace "$t" > $O/r04_clean_${tag}_trace_summary.txt
  cp gpurun_out/prof/r04_clean_$tag.json $O/r04_clean_${tag}_bench_line.json
  rm -rf gpurun_out/prof/r04_clean_$tag
done
