# usage (through gpurun): bash scripts/gpu_sweep_side.sh   -- which window-size bins run on their own stream (tuning build, 128 UHD images per step)
cd $GRAFT_REPO_ROOT
export HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so
run() { python bench.py --no-cpu-baseline --no-host-path --batch 128 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.1f img/s %.1f' % ('$1', d['ms_per_step'], d['images_per_s']))"; }
run side15
HESAFF_SIDE=0 run side0
HESAFF_SIDE=3 run side3
HESAFF_SIDE=12 run side12
HESAFF_SIDE=5 run side5
HESAFF_SIDE=10 run side10
run side15
