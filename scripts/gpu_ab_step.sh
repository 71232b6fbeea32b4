# usage (through gpurun): bash scripts/gpu_ab_step.sh <rounds> <variant> <variant> ...   (hesaff_amd/variants/<variant>.so)
# the device-resident bench step at B = 256 per variant, <rounds> interleaved rounds: step, pyramid / detection stage times, roofline fractions
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq 1 $R); do for v in "$@"; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --batch ${BATCH:-256} --steps ${STEPS:-5} --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']['serial_on_main_stream']; r=d['roofline']; e=d['roofline_detect']
print('%-10s step %7.1f ms  pyramid %6.2f  detect %6.2f  kernel frac %.3f  stage frac %.3f  extrema frac %.3f  detect stage %.0f GB/s' % ('$v', d['ms_per_step'], s['pyramid_ms'], s['detect_ms'], r['frac'], r['stage']['frac'], e['frac'], e['stage']['achieved']))"
done; done
