# usage (through gpurun): bash scripts/gpu_ab_detect.sh <variant> ... : detection stage per variant (hesaff_amd/variants/<v>.so), B = 256, two rounds interleaved
cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in "$@"; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --steps 3 --warmup 1 --fast-steps 0 --photo-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s step %.1f  extrema frac %.4f avg %.4f ms  detect %.2f ms  pyramid %.2f' % ('$v', d['ms_per_step'], d['roofline_detect']['frac'], d['roofline_detect']['avg_launch_ms'], d['stage_ms_per_step']['serial_on_main_stream']['detect_ms'], d['stage_ms_per_step']['serial_on_main_stream']['pyramid_ms']))"
done; done
