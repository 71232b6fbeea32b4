cd $GRAFT_REPO_ROOT
for v in "$@"; do
  HESAFF_PYR=$v timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pyramid or golden or end_to_end" 2>&1 | tail -1
  HESAFF_PYR=$v timeout 600 python bench.py --steps 3 --warmup 1 --batch 16 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('PYR=$v', 'achieved %.0f GB/s frac %.3f avg_launch_ms %.4f' % (r['achieved'], r['frac'], r['avg_launch_ms']), 'pyr_ms %.2f' % d['stage_ms_per_step']['pyramid_ms'], 'kp/s %.0f' % d['value'])"
done
