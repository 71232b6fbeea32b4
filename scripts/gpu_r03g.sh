# usage (through gpurun): bash scripts/gpu_r03g.sh <tag>  -- RCCL one-rank test, CLI tests, strong-scaling line with 2048 UHD images on one GPU
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "bench or cli or process_files" 2>&1 | tail -4
timeout 1200 python bench.py --scaling strong --global-images 2048 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path > gpurun_out/${TAG}_bench_strong_2048_n1.json 2> gpurun_out/${TAG}_strong.err; tail -2 gpurun_out/${TAG}_strong.err; cut -c1-900 gpurun_out/${TAG}_bench_strong_2048_n1.json
