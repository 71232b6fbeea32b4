cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in ${VARIANTS:-pyr_shfl pyr_dpp}; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --batch 128 --steps 3 --warmup 1 --fast-steps 0 --photo-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s step %.1f  kernel frac %.4f avg %.4f ms  stage %.4f  pyramid_ms %.2f detect %.2f' % ('$v', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['stage']['frac'], d['stage_ms_per_step']['serial_on_main_stream']['pyramid_ms'], d['stage_ms_per_step']['serial_on_main_stream']['detect_ms']))"
done; done
