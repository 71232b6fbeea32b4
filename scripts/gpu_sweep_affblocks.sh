cd $GRAFT_REPO_ROOT
export HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so
run() { python bench.py --no-cpu-baseline --no-host-path --batch 128 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.1f img/s %.1f' % ('$1', d['ms_per_step'], d['images_per_s']))"; }
HESAFF_AFF_BLOCKS=8 run aff8
HESAFF_AFF_BLOCKS=128 run aff128
HESAFF_AFF_BLOCKS=8 run aff8
HESAFF_AFF_BLOCKS=128 run aff128
HESAFF_AFF_BLOCKS=32 run aff32
