#!/usr/bin/env bash
# A/B of round-4 variants of the section-8d SpMV (kernels.hip.h: spmv_scs64_x) inside the CG loop `value` is quoted on:
# waves per workgroup (SB_SCS_WPB), predicated tail (SB_SCS_TAIL), unroll (SB_SCS_XU).  Same box, alternating.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
rounds=${1:-2}
for r in $(seq 1 $rounds); do
  for v in "base:" "tail:SB_SCS_TAIL=1" "u8tail:SB_SCS_TAIL=1 SB_SCS_XU=8" "wpb8:SB_SCS_WPB=8" "wpb16:SB_SCS_WPB=16" "wpb16tail:SB_SCS_WPB=16 SB_SCS_TAIL=1" "wpb16u8tail:SB_SCS_WPB=16 SB_SCS_TAIL=1 SB_SCS_XU=8" "wpb8u8tail:SB_SCS_WPB=8 SB_SCS_TAIL=1 SB_SCS_XU=8"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python3 bench.py --loops reference --no-cpu --passes clean,events --sustained-steps 0 --steps 240 --warmup 10 "${@:2}" 2>/dev/null |
      python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s %-12s %7.2f us per launch (frac %.3f)  %7.0f it/s (%.2f us per step)  preflight ok=%s' % ('$name', r['kernel'], r['avg_launch_us'], r['frac'], d['value'], 1e3*d['ms_per_step'], d['preflight']['ok']))"
  done
done
