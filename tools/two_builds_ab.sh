# same-box A/B of two builds of the library: labs/exp_a/libsbhip.so against labs/exp_b/libsbhip.so (SBHIP_LIBRARY), alternating;
# CG tests on build b first.  usage: tools/two_builds_ab.sh [extra bench args]
set -e
SBHIP_LIBRARY=$PWD/labs/exp_b/libsbhip.so python -m pytest tests/test_gpu_cg.py -x -q -m gpu -k "full_size or inside or 64 or history" > gpurun_out/two_builds_tests.log 2>&1 || { tail -20 gpurun_out/two_builds_tests.log; exit 1; }
tail -1 gpurun_out/two_builds_tests.log
for lib in a b a b a b; do SBHIP_LIBRARY=$PWD/labs/exp_$lib/libsbhip.so python bench.py --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean,events "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('build $lib: %.0f it/s (%.2f us)  SpMV %.2f us' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_us']))"; done
