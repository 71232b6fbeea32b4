# usage (through gpurun): bash scripts/gpu_r03h.sh <tag>  -- fast level 2: test, report, per-kernel times, matching score on the sequence for fast 0 / 1 / 2
cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "fast" 2>&1 | tail -6
timeout 900 python tools/fast_mode_report.py --batch 32 --level 2 > gpurun_out/${TAG}_fast2_mode.json 2> gpurun_out/${TAG}_fast2_mode.err; cut -c1-1400 gpurun_out/${TAG}_fast2_mode.json; tail -3 gpurun_out/${TAG}_fast2_mode.err
HESAFF_FAST=2 bash scripts/gpu_kernels.sh ${TAG}_fast2 32 > gpurun_out/${TAG}_kernels_serial_fast2.txt 2>&1; head -10 gpurun_out/${TAG}_kernels_serial_fast2.txt
for f in 0 1 2; do
  rm -rf /tmp/seq_${TAG}_$f; timeout 900 python tools/repeatability.py --synthetic-files /tmp/seq_${TAG}_$f --fast $f > gpurun_out/${TAG}_sequence_fast$f.json 2> gpurun_out/${TAG}_sequence_fast$f.err
  python -c "
import json; d=json.load(open('gpurun_out/${TAG}_sequence_fast$f.json'))
print('fast $f', [(e['viewpoint_deg'], round(e['repeatability'],4), round(e['matching_score'],4)) for e in d['pairs']])"
done
