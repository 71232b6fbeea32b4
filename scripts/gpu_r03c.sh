# usage (through gpurun): bash scripts/gpu_r03c.sh <tag> [natural]  -- fast mode iteration: report + per-kernel times (fast), natural-density per-kernel times (parity)
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "fast_mode" 2>&1 | tail -3
timeout 900 python tools/fast_mode_report.py --batch 32 > gpurun_out/${TAG}_fast_mode.json 2> gpurun_out/${TAG}_fast_mode.err; cut -c1-700 gpurun_out/${TAG}_fast_mode.json; tail -3 gpurun_out/${TAG}_fast_mode.err
HESAFF_FAST=1 bash scripts/gpu_kernels.sh ${TAG}_fast 32 > gpurun_out/${TAG}_kernels_serial_fast.txt 2>&1; head -12 gpurun_out/${TAG}_kernels_serial_fast.txt
if [ "$2" = "natural" ]; then
BENCH_EXTRA="--density natural" bash scripts/gpu_kernels.sh ${TAG}_nat 32 > gpurun_out/${TAG}_kernels_serial_natural.txt 2>&1; head -16 gpurun_out/${TAG}_kernels_serial_natural.txt
fi
