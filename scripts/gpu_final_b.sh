# second half of scripts/gpu_final.sh (the parts that use the tuning build): bash scripts/gpu_final_b.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=$1
bash scripts/gpu_pmc2.sh ${TAG} 8 > gpurun_out/${TAG}_pmc.md 2>&1; cat gpurun_out/${TAG}_pmc.md
bash scripts/gpu_pmc_lds.sh ${TAG} 8 > gpurun_out/${TAG}_pmc_lds.md 2>&1; cat gpurun_out/${TAG}_pmc_lds.md
bash scripts/gpu_pmc_mem.sh ${TAG} 8 > gpurun_out/${TAG}_pmc_mem.md 2>&1; cat gpurun_out/${TAG}_pmc_mem.md
bash scripts/gpu_kernels.sh ${TAG} 32 > gpurun_out/${TAG}_kernels_serial.txt 2>&1; head -30 gpurun_out/${TAG}_kernels_serial.txt
HESAFF_FAST=2 bash scripts/gpu_kernels.sh ${TAG}_fast2 32 > gpurun_out/${TAG}_kernels_serial_fast2.txt 2>&1; head -12 gpurun_out/${TAG}_kernels_serial_fast2.txt
BENCH_EXTRA="--density photo" bash scripts/gpu_kernels.sh ${TAG}_photo 32 > gpurun_out/${TAG}_kernels_serial_photo.txt 2>&1; head -12 gpurun_out/${TAG}_kernels_serial_photo.txt
