# usage (through gpurun): bash scripts/gpu_variants_ab.sh <kernel-prefix> <batch> <variant> <variant> ...  (hesaff_amd/variants/<variant>.so)
# one kernel's serial time per variant, then the overlapped step per variant, twice, interleaved
cd $GRAFT_REPO_ROOT
K=$1; BATCH=$2; shift; shift
for v in "$@"; do bash scripts/gpu_kcmp.sh $K hesaff_amd/variants/$v.so; done
for i in 1 2; do for v in "$@"; do
  HESAFF_AMD_LIB=$GRAFT_REPO_ROOT/hesaff_amd/variants/$v.so python bench.py --no-cpu-baseline --no-host-path --batch $BATCH --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s B=$BATCH step %.1f' % ('$v', d['ms_per_step']))"
done; done
