# usage: bash scripts/gpu_ab.sh VAR v1 v2 ... : full bench (batch ${BATCH:-16}) per setting
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 600 python bench.py --steps ${STEPS:-3} --warmup 1 --batch ${BATCH:-16} --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('$VAR=$v', 'kp/s %.0f img/s %.1f ms/step %.1f' % (d['value'], d['images_per_s'], d['ms_per_step']), ' '.join('%s=%.1f'%(k[:-3],v) for k,v in s.items()), 'roof %.0f'%d['roofline']['achieved'])"
done
