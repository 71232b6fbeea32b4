# usage (through gpurun): bash scripts/gpu_sweep_sched.sh   -- schedule knobs of the tuning build under the bench (batch 128)
cd $GRAFT_REPO_ROOT
export HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so
run() { python bench.py --no-cpu-baseline --no-host-path --batch 128 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.1f img/s %.1f' % ('$1', d['ms_per_step'], d['images_per_s']))"; }
run base
HESAFF_GROUP=300000 run group300k
HESAFF_GROUP=600000 run group600k
HESAFF_GROUP=2400000 run group2400k
HESAFF_AFF_BLOCKS=6 run aff6
HESAFF_AFF_BLOCKS=10 run aff10
run base
