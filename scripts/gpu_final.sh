# usage (through gpurun): bash scripts/gpu_final.sh <tag>
# round-end evidence: GPU test suite, default bench line, rocprofv3 stats + HBM traffic of the bench command, per-kernel PMC tables,
# per-kernel serial times (parity, fast, natural density), fast-mode report, config-5 sequence table
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.log 2>&1; tail -4 gpurun_out/${TAG}_tests.log
(time timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err); tail -4 gpurun_out/${TAG}_bench.err
bash scripts/gpu_profile_round.sh ${TAG}_prof > gpurun_out/${TAG}_prof.log 2>&1; tail -20 gpurun_out/${TAG}_prof.log | cut -c1-400
bash scripts/gpu_pmc2.sh ${TAG} 8 > gpurun_out/${TAG}_pmc.md 2>&1; cat gpurun_out/${TAG}_pmc.md
bash scripts/gpu_pmc_lds.sh ${TAG} 8 > gpurun_out/${TAG}_pmc_lds.md 2>&1; cat gpurun_out/${TAG}_pmc_lds.md
bash scripts/gpu_pmc_mem.sh ${TAG} 8 > gpurun_out/${TAG}_pmc_mem.md 2>&1; cat gpurun_out/${TAG}_pmc_mem.md
bash scripts/gpu_kernels.sh ${TAG} 32 > gpurun_out/${TAG}_kernels_serial.txt 2>&1; head -16 gpurun_out/${TAG}_kernels_serial.txt
HESAFF_FAST=1 bash scripts/gpu_kernels.sh ${TAG}_fast 32 > gpurun_out/${TAG}_kernels_serial_fast.txt 2>&1; head -12 gpurun_out/${TAG}_kernels_serial_fast.txt
BENCH_EXTRA="--density natural" bash scripts/gpu_kernels.sh ${TAG}_nat 32 > gpurun_out/${TAG}_kernels_serial_natural.txt 2>&1; head -12 gpurun_out/${TAG}_kernels_serial_natural.txt
timeout 900 python tools/fast_mode_report.py --batch 32 > gpurun_out/${TAG}_fast_mode.json 2> gpurun_out/${TAG}_fast_mode.err; tail -1 gpurun_out/${TAG}_fast_mode.json | cut -c1-600
rm -rf /tmp/seq_${TAG}; timeout 900 python tools/repeatability.py --synthetic-files /tmp/seq_${TAG} > gpurun_out/${TAG}_repeatability_sequence.json 2> gpurun_out/${TAG}_repeatability.err; head -c 1500 gpurun_out/${TAG}_repeatability_sequence.json
