set -e
for w in 2 1 2 1; do SB_ALPHA_WG_PER_CU=$w python bench.py --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('128^3 wg/cu', $w, d['value'], d['ms_per_step'], d['roofline_reference_layout'].get('cg_iterations_per_s'))"; done
for w in 2 1; do SB_ALPHA_WG_PER_CU=$w python bench.py --n 64 --sigma 1 --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('64^3 wg/cu', $w, d['value'], d['ms_per_step'])"; done
