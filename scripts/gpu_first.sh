set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Compute Unit" | head -4
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/first_parity.log
cat gpurun_out/first_parity.log
