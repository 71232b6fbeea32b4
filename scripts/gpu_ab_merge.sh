# usage (through gpurun): bash scripts/gpu_ab_merge.sh [batch]  - four explicit HIP streams (HESAFF_MERGE=1, the product) against eight (0), fresh processes, interleaved
cd $GRAFT_REPO_ROOT
B=${1:-256}
export HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so
for round in 1 2 3; do for m in 1 0; do
  HESAFF_MERGE=$m python bench.py --no-cpu-baseline --no-host-path --batch $B --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge=$m  B=$B step %.1f ms  %.2f M kp/s' % (d['ms_per_step'], d['value']/1e6))"
done; done
