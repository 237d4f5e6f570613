# A/B of the scalar steps inside their consumers (one GPU): default path and reference-layout loop, 128^3 and 64^3
set -e
for ab in "1 1" "0 0" "1 0" "1 1" "0 0"; do set -- $ab; python bench.py --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean --fuse-alpha $1 --fuse-beta $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_reference_layout']; print('128^3 alpha $1 beta $2: default %.0f it/s (%.2f us, %d launches)   reference layout %.0f it/s (%.2f us)' % (d['value'], 1e3*d['ms_per_step'], d['config']['launches_per_iteration'], r['cg_iterations_per_s'], 1e3*r['ms_per_step']))"; done
for ab in "1 1" "0 0"; do set -- $ab; python bench.py --n 64 --sigma 1 --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean --fuse-alpha $1 --fuse-beta $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_reference_layout']; print('64^3 alpha $1 beta $2: default %.0f it/s (%.2f us)   reference layout %.0f it/s (%.2f us)' % (d['value'], 1e3*d['ms_per_step'], r['cg_iterations_per_s'], 1e3*r['ms_per_step']))"; done
