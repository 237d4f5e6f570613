# A/B: two tiles per workgroup in the fused SpMV (second tile's header prefetched) against one (SB_FUSEP_TILES_PER_WG)
# -- needs tools/lab/two_tiles_per_workgroup_experiment.patch applied (the switch does not exist in the product)
set -e
python -m pytest tests/test_gpu_cg.py -x -q -m gpu > gpurun_out/tpw_tests.log 2>&1 || { tail -30 gpurun_out/tpw_tests.log; exit 1; }
tail -2 gpurun_out/tpw_tests.log
for t in 2 1 2 1 2 1; do SB_FUSEP_TILES_PER_WG=$t python bench.py --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean,events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('128^3 sigma 256 tiles/wg $t: %.0f it/s (%.2f us)  SpMV %.2f us' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_us']))"; done
for t in 2 1 2 1; do SB_FUSEP_TILES_PER_WG=$t python bench.py --sigma 1 --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean,events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('128^3 sigma 1 tiles/wg $t: %.0f it/s (%.2f us)  SpMV %.2f us' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_us']))"; done
