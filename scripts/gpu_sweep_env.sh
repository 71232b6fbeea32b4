# usage (through gpurun): bash scripts/gpu_sweep_env.sh <rounds> "<variant>[:ENV=val[,ENV=val...]]" ...
# the device-resident bench step per configuration (variant .so from scripts/build_variant.sh + environment of the tuning build),
# <rounds> interleaved rounds on one box: step, stream-busy times of the three concurrent stages
# BATCH / STEPS / DENSITY (dense | photo) from the environment
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq 1 $R); do for cfg in "$@"; do
  v=${cfg%%:*}; e=""; [ "$cfg" != "$v" ] && e=${cfg#*:}
  ( for kv in ${e//,/ }; do export "$kv"; done
    HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --density ${DENSITY:-dense} --batch ${BATCH:-256} --steps ${STEPS:-4} --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']['serial_on_main_stream']; q=d['stage_ms_per_step']['concurrent_stream_busy_time']
print('%-44s step %7.1f ms  pyr %5.2f det %5.2f pack %4.2f | affine %6.1f patch %6.1f sift %6.1f' % ('$cfg', d['ms_per_step'], s['pyramid_ms'], s['detect_ms'], s['pack_ms'], q['affine_ms'], q['patch_ms'], q['sift_ms']))" )
done; done
