# usage (through gpurun): bash scripts/gpu_r03a.sh <tag>
# round 3, first call: GPU test suite, default bench line (with the measured end-to-end leg), HW-queue count A/B,
# two-context overlap experiment, per-kernel serial times
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${TAG}_tests.log 2>&1; tail -14 gpurun_out/${TAG}_tests.log
(time timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err); tail -4 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["stage"]["frac"])
print("host_path", d["host_path"] and d["host_path"]["images_per_s"], "text_export", d["text_export"] and d["text_export"]["images_per_s"])
print("end_to_end", d.get("end_to_end"))
print("stages", d["stage_ms_per_step"]["serial_on_main_stream"])
PY
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 3 --warmup 1 --batch 128 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$q B=128 ms_per_step', d['ms_per_step'])"
done
B=128 timeout 600 python scripts/exp_two_ctx.py 2>&1 | tail -4
bash scripts/gpu_kernels.sh ${TAG} 32 > gpurun_out/${TAG}_kernels_serial.txt 2>&1; head -20 gpurun_out/${TAG}_kernels_serial.txt
