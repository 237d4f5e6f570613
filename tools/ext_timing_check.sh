# how the SpMV launch is timed inside the loop: {hipExtLaunchKernelGGL stamps, hipEventRecord brackets} x {events without / with the system fence}
set -e
for spec in "ext:0x20000000" "ext:0" "record:0x20000000" "record:0"; do
  how=${spec%%:*}; fl=${spec#*:}
  SB_SPMV_TIMING=$how SB_EVENT_FLAGS=$fl python bench.py --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean,events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; q=d['roofline_reference_layout']; print('$how flags $fl: %s %.2f us   %s %.2f us (frac %.3f)   clean %.0f it/s' % (r['kernel'], r['avg_launch_us'] if 'avg_launch_us' in r else r.get('launch_us',0), q['kernel'], q.get('avg_launch_us', q.get('launch_us',0)), q['frac'], d['value']))"
done
