# usage (through gpurun): bash scripts/gpu_variant_bench.sh <variant> ...   -- serial stage times of the bench under each hesaff_amd/variants/<variant>.so
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --batch 128 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['stage_ms_per_step']['serial_on_main_stream']; print('%-8s step %.1f pyr %.2f det %.2f pack %.2f kernel frac %.3f stage frac %.3f' % ('$v', d['ms_per_step'], s['pyramid_ms'], s['detect_ms'], s['pack_ms'], r['frac'], r['stage']['frac']))"
done
