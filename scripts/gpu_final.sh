# usage (through gpurun): bash scripts/gpu_final.sh <tag>
# round-end evidence, ONE pass (needs the tuning library: `make -C hesaff_amd/csrc tuning` before the gpurun call): GPU test suite, default bench line,
# rocprofv3 stats + HBM traffic of the bench command, per-kernel PMC tables (dense and photographs), counters of the overlapped step,
# per-kernel serial times (parity, fast = 2, photographs), fast-mode reports, config-5 sequence tables (band noise and photograph, fast 0 / 2),
# the file path inside 2 / 4 / 8 CPUs with per-thread CPU seconds, end-to-end thread sweep, JPEG lists (UHD and 1024x768; pixels on the device / on the host)
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.log 2>&1; tail -4 gpurun_out/${TAG}_tests.log
T0=$(date +%s); timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench.py wall seconds: $(( $(date +%s) - T0 ))" | tee -a gpurun_out/${TAG}_bench.err
bash scripts/gpu_profile_round.sh ${TAG}_prof > gpurun_out/${TAG}_prof.log 2>&1; tail -20 gpurun_out/${TAG}_prof.log | cut -c1-400
bash scripts/gpu_pmc2.sh ${TAG} 8 > gpurun_out/${TAG}_pmc.md 2>&1; cat gpurun_out/${TAG}_pmc.md
bash scripts/gpu_pmc_lds.sh ${TAG} 8 > gpurun_out/${TAG}_pmc_lds.md 2>&1; cat gpurun_out/${TAG}_pmc_lds.md
bash scripts/gpu_pmc_mem.sh ${TAG} 8 > gpurun_out/${TAG}_pmc_mem.md 2>&1; cat gpurun_out/${TAG}_pmc_mem.md
# the same three tables on the photograph mosaics (k_patch_large_rows / _finish are the second largest kernels there), and the counters of the overlapped step
for S in pmc2 pmc_lds pmc_mem; do BENCH_EXTRA="--density photo" PMC_ROWS=14 bash scripts/gpu_$S.sh ${TAG}_photo 8 > gpurun_out/${TAG}_${S}_photo.md 2>&1; done; cat gpurun_out/${TAG}_pmc2_photo.md
bash scripts/gpu_pmc_overlap.sh ${TAG} 256 > gpurun_out/${TAG}_pmc_overlap.log 2>&1; head -30 gpurun_out/stage_util_${TAG}.md | cut -c1-300
bash scripts/gpu_kernels.sh ${TAG} 32 > gpurun_out/${TAG}_kernels_serial.txt 2>&1; head -30 gpurun_out/${TAG}_kernels_serial.txt
HESAFF_FAST=2 bash scripts/gpu_kernels.sh ${TAG}_fast2 32 > gpurun_out/${TAG}_kernels_serial_fast2.txt 2>&1; head -12 gpurun_out/${TAG}_kernels_serial_fast2.txt
BENCH_EXTRA="--density photo" bash scripts/gpu_kernels.sh ${TAG}_photo 32 > gpurun_out/${TAG}_kernels_serial_photo.txt 2>&1; head -12 gpurun_out/${TAG}_kernels_serial_photo.txt
timeout 900 python tools/fast_mode_report.py --batch 32 > gpurun_out/${TAG}_fast_mode.json 2> gpurun_out/${TAG}_fast_mode.err; tail -1 gpurun_out/${TAG}_fast_mode.json | cut -c1-600
timeout 900 python tools/fast_mode_report.py --batch 32 --photo > gpurun_out/${TAG}_fast_mode_photo.json 2>> gpurun_out/${TAG}_fast_mode.err; tail -1 gpurun_out/${TAG}_fast_mode_photo.json | cut -c1-600
for f in 0 2; do
  rm -rf /tmp/seq_${TAG}; timeout 900 python tools/repeatability.py --synthetic-files /tmp/seq_${TAG} --fast $f > gpurun_out/${TAG}_repeatability_sequence_fast$f.json 2> gpurun_out/${TAG}_repeatability.err
  for ph in 0 1; do rm -rf /tmp/seqp_${TAG}; timeout 900 python tools/repeatability.py --synthetic-files /tmp/seqp_${TAG} --photo $ph --fast $f > gpurun_out/${TAG}_repeatability_photo${ph}_fast$f.json 2>> gpurun_out/${TAG}_repeatability.err; done
done
python - <<PY
import json, glob
for q in sorted(glob.glob("gpurun_out/${TAG}_repeatability_*.json")):
    try:
        d = json.load(open(q))
        print(q.rsplit("/", 1)[1], [(e.get("viewpoint_deg"), round(e["repeatability"], 4), round(e["matching_score"], 4)) for e in d["pairs"]])
    except Exception as e:
        print(q, "unreadable", e)
PY
timeout 900 python scripts/e2e_budget_probe.py 512 2 4 8 > gpurun_out/${TAG}_e2e_budget_probe.txt 2>&1; grep -v "amdgpu.ids" gpurun_out/${TAG}_e2e_budget_probe.txt | cut -c1-200
timeout 900 python scripts/e2e_thread_sweep.py 384 > gpurun_out/${TAG}_e2e_thread_sweep.txt 2>&1; tail -12 gpurun_out/${TAG}_e2e_thread_sweep.txt | cut -c1-300
timeout 600 python scripts/jpeg_list_rate.py 512 > gpurun_out/${TAG}_jpeg_list_rate.txt 2>&1; timeout 600 python scripts/jpeg_list_rate.py 2048 1024 768 >> gpurun_out/${TAG}_jpeg_list_rate.txt 2>&1; cut -c1-220 gpurun_out/${TAG}_jpeg_list_rate.txt
