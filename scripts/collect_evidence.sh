# usage (here, after gpurun merged gpurun_out/): bash scripts/collect_evidence.sh <tag of scripts/gpu_final.sh> <round, e.g. r04>
# copies the summaries of one evidence pass into profiles/ under the round's names
set -e
cd "$(dirname "$0")/.."
T=$1; R=$2; G=gpurun_out
cp $G/${T}_bench.json profiles/${R}_bench_full.json
cp $G/${T}_prof/stats/p_kernel_stats.csv profiles/${R}_kernel_stats_bench.csv
cp $G/${T}_prof/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $G/${T}_prof/pmc_traffic_raw.json profiles/${R}_pmc_traffic.json
cp $G/${T}_prof/pmc_traffic_raw.json profiles/pmc_traffic.json
(echo "# Per-kernel counters, ${R} (every kernel alone on the device, batch 8; scripts/gpu_pmc2.sh, gpu_pmc_lds.sh, gpu_pmc_mem.sh)"; echo; cat $G/${T}_pmc.md; echo; cat $G/${T}_pmc_lds.md; echo; cat $G/${T}_pmc_mem.md) > profiles/${R}_pmc_kernels.md
(echo "# Per-kernel counters on the photograph mosaics, ${R} (every kernel alone on the device, batch 8, --density photo)"; echo; cat $G/${T}_pmc2_photo.md; echo; cat $G/${T}_pmc_lds_photo.md; echo; cat $G/${T}_pmc_mem_photo.md) > profiles/${R}_pmc_kernels_photographs.md
cp $G/stage_util_${T}.md profiles/${R}_stage_utilisation.md
cp $G/${T}_kernels_serial.txt profiles/${R}_kernels_serial.txt
cp $G/${T}_kernels_serial_fast2.txt profiles/${R}_kernels_serial_fast2.txt
cp $G/${T}_kernels_serial_photo.txt profiles/${R}_kernels_serial_photographs.txt
cp $G/${T}_fast_mode.json profiles/${R}_fast_mode.json
cp $G/${T}_fast_mode_photo.json profiles/${R}_fast_mode_photographs.json
grep -v '^{"metric' $G/${T}_e2e_thread_sweep.txt > profiles/${R}_e2e_thread_sweep.txt
grep -v amdgpu.ids $G/${T}_e2e_budget_probe.txt > profiles/${R}_e2e_budget_probe.txt
tail -4 $G/${T}_tests.log > profiles/${R}_gpu_tests.log
grep -v amdgpu.ids $G/${T}_jpeg_list_rate.txt > profiles/${R}_jpeg_list_rate.txt
python3 - $T $R <<'PY'
import json, glob, sys
t, r = sys.argv[1], sys.argv[2]
out = {}
for q in sorted(glob.glob('gpurun_out/%s_repeatability_*.json' % t)):
    d = json.load(open(q)); name = q.rsplit('%s_repeatability_' % t, 1)[1][:-5]
    out[name] = {"data": d["data"], "command": d["command"], "fast": d["fast"], "pairs": d["pairs"]}
json.dump(out, open('profiles/%s_repeatability_sequences.json' % r, 'w'), indent=1)
print(sorted(out))
PY
