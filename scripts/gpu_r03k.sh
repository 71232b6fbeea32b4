# usage (through gpurun): bash scripts/gpu_r03k.sh <tag>  -- full GPU suite + per-kernel serial times + step at B = 128 / 256
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; tail -4 gpurun_out/${TAG}_tests.log
bash scripts/gpu_kernels.sh ${TAG} 32 > gpurun_out/${TAG}_kernels_serial.txt 2>&1; head -14 gpurun_out/${TAG}_kernels_serial.txt
for i in 1 2; do timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=256 ms_per_step', d['ms_per_step'], 'value', d['value'])"; done
