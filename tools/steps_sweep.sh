# CG it/s as a function of the length of the timed run (device clock state: DESIGN 7), default path and reference-layout loop
for k in 20 480 5000 30000 20; do python3 bench.py --no-cpu --steps $k --warmup 5 --no-preflight --passes clean 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_reference_layout']; print('steps %6d: default %.0f it/s (%.2f us)   reference layout %.0f it/s (%.2f us)' % ($k, d['value'], 1e3*d['ms_per_step'], r['cg_iterations_per_s'], 1e3*r['ms_per_step']))"; done
