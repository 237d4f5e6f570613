#!/usr/bin/env bash
# What are the device's clocks while the reference-layout CG loop runs -- plain, and under rocprofv3 --kernel-trace?
# (the loop runs 13 % faster under the profiler on some boxes: profiles/README.md).  Samples rocm-smi a few times per run.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
sample() { for i in 1 2 3 4 5 6; do sleep 0.7; /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | tr -s ' ' | tr '\n' ';'; echo; done; }
echo "== idle"; /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | tr -s ' ' | tr '\n' ';'; echo
echo "== plain"
python3 bench.py --no-cpu --steps 30000 --warmup 5 --no-preflight --passes clean --pack-mode 0 > gpurun_out/clock_plain.json 2>/dev/null &
pid=$!; sleep 6; sample; wait $pid
python3 -c "import json; d=json.load(open('gpurun_out/clock_plain.json')); print('plain: %.0f it/s' % d['value'])"
echo "== under rocprofv3 --kernel-trace"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/clockwatch -o r1 -- python3 bench.py --no-cpu --steps 30000 --warmup 5 --no-preflight --passes clean --pack-mode 0 > gpurun_out/clock_prof.json 2>/dev/null &
pid=$!; sleep 8; sample; wait $pid
python3 -c "import json; d=json.load(open('gpurun_out/clock_prof.json')); print('under the profiler: %.0f it/s' % d['value'])"
rm -rf gpurun_out/prof/clockwatch
