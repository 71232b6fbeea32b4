# usage: bash scripts/gpu_sweep.sh VAR v1 v2 ...   -> roofline + pyramid ms per setting
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 600 python bench.py --steps ${STEPS:-2} --warmup 1 --batch ${BATCH:-16} --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('$VAR=$v', 'achieved %.0f GB/s frac %.3f avg_launch_ms %.4f' % (r['achieved'], r['frac'], r['avg_launch_ms']), 'pyr_ms %.2f' % d['stage_ms_per_step']['pyramid_ms'], 'kp/s %.0f' % d['value'])
"
done
