# usage (through gpurun): bash scripts/gpu_env_ab.sh <batch> "<ENV=..>" "<ENV=..>" ...  -- step time of the tuning library under each environment, twice, interleaved
cd $GRAFT_REPO_ROOT
BATCH=$1; shift
for i in 1 2; do for spec in "$@"; do
  env $spec HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so python bench.py --no-cpu-baseline --no-host-path --batch $BATCH --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s B=$BATCH step %.1f' % ('$spec', d['ms_per_step']))"
done; done
