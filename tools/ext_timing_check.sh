set -e
mkdir -p gpurun_out/ext
python bench.py --steps 60 --warmup 10 --no-cpu > gpurun_out/ext/ext.json 2> gpurun_out/ext/ext.err
SB_SPMV_TIMING=record python bench.py --steps 60 --warmup 10 --no-cpu > gpurun_out/ext/record.json 2> gpurun_out/ext/record.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ext/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu --no-preflight --passes clean,events > $GRAFT_REPO_ROOT/gpurun_out/ext/prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/ext/prof.err
cd $GRAFT_REPO_ROOT
for f in ext record prof; do python tools/show_bench.py gpurun_out/ext/$f.json | grep -E "it/s|roofline|reference"; done
find gpurun_out/ext/prof -name "*kernel_stats.csv" | head -1 | xargs head -8
