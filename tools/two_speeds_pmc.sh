#!/usr/bin/env bash
# usage (on the GPU box, from the repo root): tools/two_speeds_pmc.sh [rounds=2]
# what decides the two speeds (~142 / ~155 us per step; spmv_scs64 ~117 / ~128 us) of the section-8d loop?
# Fresh processes, alternating: a plain clean timing, then the same program directly behind `rocprofv3 --kernel-trace --pmc <one
# counter group> --` (kernel trace + counters only: what gpurun allows).  Every profiled process tells its own kind by its
# spmv_scs64 duration; tools/two_speeds_report.py tabulates duration against counters per process.
# `tools/two_speeds_pmc.sh auto [nplain=8]`: nplain plain processes first; the counter rounds only if this box shows both kinds
# (spread of the plain step times > 4 %) -- most boxes of the pool show one kind only.
rounds=${1:-2}
nplain=${2:-8}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/r04/two_speeds
mkdir -p $out
/opt/rocm/bin/rocm-smi --showclocks --showpower --showmeminfo vram > $out/smi_idle.txt 2>&1
declare -A SETS
SETS[sq]="GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
SETS[utcl1]="GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum"
SETS[tcp]="GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
SETS[tcc]="GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
SETS[ea]="GRBM_GUI_ACTIVE TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_sum"
SETS[chan]="GRBM_GUI_ACTIVE TCC_EA0_RDREQ TCC_TAG_STALL"
if [ "$rounds" = auto ]; then
  for j in $(seq 1 $nplain); do python3 tools/two_speeds_probe.py scout_$j 2>> $out/plain.err | grep two_speeds_probe >> $out/scout.txt; done
  cat $out/scout.txt
  spread=$(python3 -c "
import re,sys
v=[float(re.search(r'best segment ([0-9.]+)', l).group(1)) for l in open('$out/scout.txt') if 'best segment' in l]
print('%.3f' % (max(v)/min(v)-1.0))")
  echo "scout: spread of the plain step times on this box: $spread"
  if python3 -c "import sys; sys.exit(0 if $spread > 0.04 else 1)"; then rounds=3; else echo "one kind only on this box: no counter rounds"; exit 0; fi
  # both kinds on this box: is it where the buffers land?  (one process, allocation sequence varied)
  for j in 1 2 3; do python3 tools/two_speeds_inproc.py 2>> $out/plain.err | grep two_speeds_inproc | tee -a $out/inproc.txt; echo "-- next process" | tee -a $out/inproc.txt; done
fi
i=0
for r in $(seq 1 $rounds); do
  for s in sq utcl1 tcp tcc ea chan; do
    i=$((i+1))
    python3 tools/two_speeds_probe.py plain_$i >> $out/plain.txt 2>> $out/plain.err
    rocprofv3 --kernel-trace --pmc ${SETS[$s]} --output-format csv -d $out/p${i}_$s -o r1 -- python3 tools/two_speeds_probe.py pmc_${i}_$s 2 > $out/p${i}_$s.log 2>&1 \
      || echo "pass $i ($s) failed: $(tail -2 $out/p${i}_$s.log | tr '\n' ' ')"
    echo "process pair $i ($s) done: $(tail -1 $out/plain.txt)"
  done
done
python3 tools/two_speeds_report.py $out | tee $out/report.txt
# keep the merge-back small: the per-launch csv files are condensed in report.txt
find $out -name "*_counter_collection.csv" -size +4M -delete
find $out -name "*_kernel_trace.csv" -size +4M -delete
