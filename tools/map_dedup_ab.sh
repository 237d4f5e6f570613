# A/B: distinct slot maps stored once (default) against one map per tile (SB_PACK_MAP_DEDUP=0); sigma > 1 only (mapped windows)
set -e
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cg.py -x -q -m gpu > gpurun_out/dedup_tests.log 2>&1 || { tail -30 gpurun_out/dedup_tests.log; exit 1; }
tail -2 gpurun_out/dedup_tests.log
SB_PACK_REPORT=1 python bench.py --steps 20 --warmup 5 --no-cpu --no-preflight --passes clean 2>&1 >/dev/null | grep "slot maps" | head -3
for t in 1 0 1 0 1 0; do SB_PACK_MAP_DEDUP=$t python bench.py --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean,events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('128^3 sigma 256 dedup $t: %.0f it/s (%.2f us)  SpMV %.2f us  moves %.1f MB' % (d['value'], 1e3*d['ms_per_step'], r['avg_launch_us'], r['bytes_per_launch']/1e6))"; done
for t in 1 0; do SB_PACK_MAP_DEDUP=$t python bench.py --fuse-p 0 --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean,events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('128^3 sigma 256 separate p update, dedup $t: %.0f it/s  SpMV %.2f us  moves %.1f MB' % (d['value'], r['avg_launch_us'], r['bytes_per_launch']/1e6))"; done
