# usage (through gpurun): bash scripts/gpu_ab_libs.sh <libA> <libB> [batch]  -- A/B of two tuning libraries in one call:
# per-kernel serial times (32 images) and the overlapped step (batch images, alternating A B A B)
cd $GRAFT_REPO_ROOT
A=$1; B=$2; BATCH=${3:-256}
for L in $A $B; do
  bash scripts/gpu_kernels.sh ab_$(basename $L .so) 32 $GRAFT_REPO_ROOT/$L 2>&1 | head -13
done
for i in 1 2; do for L in $A $B; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/$L timeout 600 python bench.py --steps 3 --warmup 1 --batch $BATCH --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L B=$BATCH ms_per_step %.1f' % d['ms_per_step'])"
done; done
