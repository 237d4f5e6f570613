#!/usr/bin/env bash
# A/B of the window staging of spmv_prog_fusep (pack.hip.h: PAIR): SB_FUSEP_PAIR=1 two slots per thread and step with 16-byte
# loads (round 4), SB_FUSEP_PAIR=0 round 3's one slot per thread and step.  Same box, alternating; the structure-exploiting loop
# only (bench.py --loops structure).  usage: tools/fusep_pair_ab.sh [rounds=3]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
rounds=${1:-3}
run() { # label, extra bench args...
  local label=$1; shift
  for r in $(seq 1 $rounds); do for p in 1 0; do
    SB_FUSEP_PAIR=$p python3 bench.py --loops structure --no-cpu --no-preflight --passes clean,events --sustained-steps 0 --steps 240 --warmup 10 "$@" 2>/dev/null |
      python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-22s pair %s: %-16s %7.2f us per launch  %8.0f it/s (%.2f us per step)' % ('$label', '$p', r['kernel'], r['avg_launch_us'], d['value'], 1e3*d['ms_per_step']))"
  done; done
}
run "128^3 sigma 256"
run "128^3 sigma 1" --sigma 1
run "128^3 crs mirror" --fmt crs
run "64^3 sigma 1" --n 64 --sigma 1
run "256^3 sigma 256" --n 256 --steps 60
