# usage (through gpurun): bash scripts/gpu_r03b.sh <tag>   -- fast-mode fused descriptor: report, per-kernel times of both modes
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "fast_mode or process_files or detect_batch_cb" 2>&1 | tail -3
timeout 900 python tools/fast_mode_report.py --batch 32 > gpurun_out/${TAG}_fast_mode.json 2> gpurun_out/${TAG}_fast_mode.err; cut -c1-900 gpurun_out/${TAG}_fast_mode.json; tail -3 gpurun_out/${TAG}_fast_mode.err
bash scripts/gpu_kernels.sh ${TAG} 32 > gpurun_out/${TAG}_kernels_serial.txt 2>&1; head -18 gpurun_out/${TAG}_kernels_serial.txt
HESAFF_FAST=1 bash scripts/gpu_kernels.sh ${TAG}_fast 32 > gpurun_out/${TAG}_kernels_serial_fast.txt 2>&1; head -14 gpurun_out/${TAG}_kernels_serial_fast.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "host_path", d["host_path"]["images_per_s"], "text_export", d["text_export"]["images_per_s"])
e = d["end_to_end"]; print("end_to_end", e["images"], e["images_per_s"], e.get("fraction_of_host_path"))
print("natural", d["natural_density"]); print("probe", d["hbm_probe"]); print("detect", d["roofline_detect"])
PY
