#!/usr/bin/env bash
# usage (on the GPU box, from the repo root): tools/prof_run.sh <tag> <python-script-and-args...>
# Four rocprofv3 runs of the SAME command, each with --kernel-trace/--stats or ONE --pmc group only (gpurun refuses
# --pmc combined with the heavier trace domains): kt (kernel trace + stats), fetch (FETCH_SIZE), write (WRITE_SIZE),
# sq (SQ wave / instruction counters), tcc (L2 hit / miss, memory-side requests).  Output: gpurun_out/prof/<tag>/*;
# condense with tools/summarize_prof2.py into profiles/.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/prof/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o r1 -- python3 "$@" > $out/kt.log 2>&1 || echo "kt pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o r1 -- python3 "$@" > $out/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o r1 -- python3 "$@" > $out/write.log 2>&1 || echo "write pass failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD \
  --output-format csv -d $out/sq -o r1 -- python3 "$@" > $out/sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum \
  --output-format csv -d $out/tcc -o r1 -- python3 "$@" > $out/tcc.log 2>&1 || echo "tcc pass failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS \
  --output-format csv -d $out/sq2 -o r1 -- python3 "$@" > $out/sq2.log 2>&1 || echo "sq2 pass failed"
find $out -name "*.csv" | head -20
