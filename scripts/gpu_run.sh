# usage: bash scripts/gpu_run.sh <tag> [tests] [bench] [prof]   (runs on the GPU box via gpurun)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out
for what in "$@"; do
case $what in
tests)
  timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -25 > gpurun_out/${TAG}_tests.log; cat gpurun_out/${TAG}_tests.log;;
bench)
  timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  cat gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err;;
benchq)
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  cat gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err;;
prof)
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1)
  tail -2 gpurun_out/${TAG}_prof.log
  f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); echo $f; head -25 $f | cut -c1-200;;
esac
done
