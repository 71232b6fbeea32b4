# usage (through gpurun): bash scripts/gpu_round.sh <tag> [tests|notests] [bench args...]
# GPU test suite + one bench line; logs under gpurun_out/<tag>_*
cd $GRAFT_REPO_ROOT
TAG=$1; shift
MODE=${1:-tests}; shift
mkdir -p gpurun_out
if [ "$MODE" = "tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
  echo "tests exit $?" >> gpurun_out/${TAG}_tests.log
  tail -15 gpurun_out/${TAG}_tests.log
fi
timeout 900 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit $?"
head -c 2500 gpurun_out/${TAG}_bench.json
tail -5 gpurun_out/${TAG}_bench.err
